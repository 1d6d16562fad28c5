#!/bin/bash
# round 6, session k: text encoder on its own stream: grounding tests + step A/B
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_imgconv.py tests/test_gpu_resnet2d.py -m gpu -q -s -x -k 'stem or resnet' > $OUT/r6k_stem_tests.txt 2>&1; echo "rc $?"; tail -3 $OUT/r6k_stem_tests.txt; grep 'stem + pool' $OUT/r6k_stem_tests.txt
timeout 900 python -m pytest tests/test_gpu_config4.py tests/test_gpu_grounding.py tests/test_gpu_optim_table.py tests/test_gpu_insitu.py -m gpu -q -x > $OUT/r6k_tests.txt 2>&1; echo "rc $?"; tail -3 $OUT/r6k_tests.txt
B="python bench.py --no-cpu-baseline --only grounding --steps 10 --warmup 3 --other-steps 10"
for rep in 1 2; do
  for v in "ES_TEXT_ASYNC=1" "ES_TEXT_ASYNC=0" "ES_TEXT_ASYNC=1 ES_STEM_POOL=0"; do
    env $v timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['value'], d.get('parity'))" | tee -a $OUT/r6k_ab.txt
  done
done
B="python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 5"
for rep in 1 2; do
  for v in "ES_STEM_POOL=1" "ES_STEM_POOL=0"; do
    env $v timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mv3ddet $v', d['ms_per_step'], d['value'])" | tee -a $OUT/r6k_ab.txt
  done
done
