#!/bin/bash
# round 6, session p: 64-channel stem + pool (occupancy): tests + step A/B
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_imgconv.py tests/test_gpu_resnet2d.py tests/test_gpu_config5.py tests/test_gpu_occ.py tests/test_gpu_linear.py -m gpu -q -s -x > $OUT/r6p_tests.txt 2>&1; echo "rc $?"; grep 'stem + pool\|passed\|failed' $OUT/r6p_tests.txt | cut -c1-200
B="python bench.py --no-cpu-baseline --only occupancy --steps 10 --warmup 3 --other-steps 10"
for rep in 1 2 3; do
  for v in "ES_STEM_POOL=1" "ES_STEM_POOL=0"; do
    env $v timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('occupancy $v', d['ms_per_step'], d['value'])" | tee -a $OUT/r6p_ab.txt
  done
done
