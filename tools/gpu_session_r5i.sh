#!/bin/bash
# round 5, session i: the default bench line (all configurations + the from-files leg + CPU baselines), the four-stream timeline /
# dependent chain of mv-3ddet with the hardware queues told apart
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
B="$GRAFT_REPO_ROOT/bench.py"
db () { find /tmp/prof_$1 -name '*.db' | head -1; }
timeout 900 python bench.py > $OUT/r5i_bench_default.json 2> $OUT/r5i_bench_default.err; echo "rc $?"
C1="python $B --no-cpu-baseline --no-other-configs --steps 4 --warmup 2"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt_det -o p -- $C1 > /tmp/prof_kt_det.log 2>&1); echo "rc $?"
python tools/rocpd_critical.py "$(db kt_det)" > $OUT/r5i_critical_chain.txt 2>&1
python tools/rocpd_timeline.py "$(db kt_det)" > $OUT/r5i_stream_timeline.txt 2>&1
python tools/rocpd_stats.py "$(db kt_det)" $OUT/r5i_kernel_stats.txt > /dev/null
tail -3 $OUT/r5i_bench_default.err
