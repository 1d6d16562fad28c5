#!/bin/bash
# round 5, session j: projection meta built early (mv-3ddet / grounding step), default line incl. the from-files leg
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 300 python bench.py --no-cpu-baseline --no-other-configs --steps 30 --warmup 5 > $OUT/r5j_bench_mv3ddet.json 2> $OUT/r5j_err1.txt; echo "rc $?"
timeout 400 python bench.py --no-cpu-baseline --only grounding --steps 20 --other-steps 20 --warmup 3 > $OUT/r5j_bench_grounding.json 2> $OUT/r5j_err2.txt; echo "rc $?"
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/r5j_bench_default.json 2> $OUT/r5j_bench_default.err; echo "rc $?"
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_fusion_losses.py tests/test_gpu_grounding.py -x -q > $OUT/r5j_tests.txt 2>&1; echo "rc $?"
tail -4 $OUT/r5j_tests.txt
