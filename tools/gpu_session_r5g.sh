#!/bin/bash
# round 5, session g: collector freeze reaching every detector, 3x3 image weight gradients per kernel row (A/B), neck tests
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_dconv.py tests/test_gpu_resnet2d.py -x -q > $OUT/r5g_tests_a.txt 2>&1; echo "rc $?"
ES_BENCH_DIAG=1 timeout 500 python bench.py --no-cpu-baseline --only grounding --steps 40 --other-steps 40 --warmup 3 > $OUT/r5g_bench_grounding_diag.json 2> $OUT/r5g_bench_grounding_diag.err; echo "rc $?"
ES_IMG_WGRAD=0 ES_BENCH_DUMP=$OUT/r5g_launches_img0.jsonl timeout 300 python bench.py --no-cpu-baseline --no-other-configs --steps 21 --warmup 5 > $OUT/r5g_bench_mv3ddet_img0.json 2> $OUT/r5g_err0.txt; echo "rc $?"
ES_IMG_WGRAD=1 ES_BENCH_DUMP=$OUT/r5g_launches_img1.jsonl timeout 300 python bench.py --no-cpu-baseline --no-other-configs --steps 21 --warmup 5 > $OUT/r5g_bench_mv3ddet_img1.json 2> $OUT/r5g_err1.txt; echo "rc $?"
tail -4 $OUT/r5g_tests_a.txt
