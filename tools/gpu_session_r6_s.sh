#!/bin/bash
# round 6, session s: which stream carries the text encoder -- the grounding leg inside the default run and alone
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_config4.py tests/test_gpu_grounding.py tests/test_gpu_optim_table.py -m gpu -q -x > $OUT/r6s_tests.txt 2>&1; echo "rc $?"; tail -2 $OUT/r6s_tests.txt
for rep in 1 2; do
  for v in "ES_TEXT_STREAM=wgrad" "ES_TEXT_STREAM=own" "ES_TEXT_ASYNC=0"; do
    env $v timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); o=d['other_configs']; print('default run, $v: mv3ddet', d['ms_per_step'], 'from_files', o['from_files']['ms_per_step'], 'grounding', o['grounding']['ms_per_step'], 'occupancy', o['occupancy']['ms_per_step'])" | tee -a $OUT/r6s_ab.txt
  done
done
B="python bench.py --no-cpu-baseline --only grounding --steps 10 --warmup 3 --other-steps 10"
for rep in 1 2 3; do
  for v in "ES_TEXT_STREAM=wgrad" "ES_TEXT_STREAM=own"; do
    env $v timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('only grounding, $v', d['ms_per_step'])" | tee -a $OUT/r6s_ab.txt
  done
done
