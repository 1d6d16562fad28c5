#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
B="python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 5"
for q in 4 8; do
  for pf in 0 1; do
    GPU_MAX_HW_QUEUES=$q ES_NEXT_PREFETCH=$pf timeout 200 $B > $OUT/r4e_bench_q${q}_pf${pf}.json 2> $OUT/r4e_bench_q${q}_pf${pf}.err; echo "rc $?"
  done
done
python - <<'PY'
import json
for q in (4,8):
  for pf in (0,1):
    try:
        d=json.load(open(f'gpurun_out/r4e_bench_q{q}_pf{pf}.json')); print('queues',q,'prefetch',pf, d['ms_per_step'], d['value'])
    except Exception as e: print(q,pf,'ERR', e)
PY
db () { find /tmp/prof_$1 -name '*.db' | head -1; }
CMD="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-other-configs --steps 4 --warmup 3"
(cd /tmp && GPU_MAX_HW_QUEUES=4 timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/prof_ks -o p -- $CMD > /tmp/prof_ks.log 2>&1); echo "rc $?"
python tools/rocpd_timeline.py "$(db ks)" 6 > $OUT/r4e_timeline.txt 2>&1
head -45 $OUT/r4e_timeline.txt
timeout 900 python -m pytest tests/test_gpu_grounding.py tests/test_gpu_insitu.py tests/test_gpu_prefetch.py -q -s -x -p no:cacheprovider > $OUT/r4e_tests.txt 2>&1
echo "pytest rc $?" >> $OUT/r4e_tests.txt
grep -v Warning $OUT/r4e_tests.txt | tail -40
