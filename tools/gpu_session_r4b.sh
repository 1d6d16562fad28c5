#!/bin/bash
# round 4, session b: next-batch prefetch -- parity test, A/B of the step, stream timeline
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 400 python -m pytest tests/test_gpu_prefetch.py tests/test_gpu_experimental.py tests/test_gpu_model.py -q -s -x -p no:cacheprovider > $OUT/r4b_tests.txt 2>&1
echo "pytest rc $?" >> $OUT/r4b_tests.txt
tail -15 $OUT/r4b_tests.txt
B="python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 5"
ES_NEXT_PREFETCH=0 timeout 200 $B > $OUT/r4b_bench_pf0.json 2> $OUT/r4b_bench_pf0.err; echo "rc $?"
timeout 200 $B > $OUT/r4b_bench_pf1.json 2> $OUT/r4b_bench_pf1.err; echo "rc $?"; tail -3 $OUT/r4b_bench_pf1.err
python - <<'PY'
import json
for t in ('pf0','pf1'):
    try:
        d=json.load(open(f'gpurun_out/r4b_bench_{t}.json')); print(t, d['ms_per_step'], d['value'], d['roofline']['launches_per_step'], d.get('hipgraph'))
    except Exception as e: print(t, 'ERR', e)
PY
db () { find /tmp/prof_$1 -name '*.db' | head -1; }
CMD="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-other-configs --steps 4 --warmup 3"
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/prof_ks -o p -- $CMD > /tmp/prof_ks.log 2>&1); echo "rc $?"
python tools/rocpd_stats.py "$(db ks)" $OUT/r4b_kernel_stats.txt > /dev/null
python tools/rocpd_timeline.py "$(db ks)" 6 > $OUT/r4b_timeline.txt 2>&1
head -50 $OUT/r4b_timeline.txt
