#!/bin/bash
# closing session of round 3: full GPU suite + smoke, the default bench line, kernel traces of the final code
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 620 python -m pytest tests -m gpu -q -s -x -p no:cacheprovider > $OUT/r3_gputest_full.txt 2>&1
echo "pytest rc $?" >> $OUT/r3_gputest_full.txt
tail -5 $OUT/r3_gputest_full.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/r3_smoke.txt 2>&1
echo "smoke rc $?"; tail -2 $OUT/r3_smoke.txt
timeout 400 python bench.py > $OUT/r3_bench_default.json 2> $OUT/r3_bench_default.err
echo "bench rc $?"
tail -c 300 $OUT/r3_bench_default.err
head -c 400 $OUT/r3_bench_default.json
B="$GRAFT_REPO_ROOT/bench.py"
db () { find /tmp/prof_$1 -name '*.db' | head -1; }
CMD="python $B --no-cpu-baseline --no-other-configs --steps 4 --warmup 3"
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/prof_ks -o p -- $CMD > /tmp/prof_ks.log 2>&1); echo "rc $?"
python tools/rocpd_stats.py "$(db ks)" $OUT/r3_kernel_stats.txt > /dev/null
python tools/rocpd_timeline.py "$(db ks)" 4 > $OUT/r3_stream_timeline.txt 2>&1
(cd /tmp && ES_TWO_STREAMS=0 ES_WGRAD_ASYNC=0 timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/prof_ss -o p -- $CMD > /tmp/prof_ss.log 2>&1); echo "rc $?"
python tools/rocpd_stats.py "$(db ss)" $OUT/r3_single_stream_kernel_stats.txt > /dev/null
ls -la $OUT | tail -8
