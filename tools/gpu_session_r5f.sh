#!/bin/bash
# round 5, session f: 3x3 image weight gradients in one workgroup (parity + step A/B), collector log of the grounding loop, from-files leg
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_dconv.py tests/test_gpu_resnet2d.py -x -q > $OUT/r5f_tests_a.txt 2>&1; echo "rc $?"
ES_IMG_WGRAD=0 timeout 300 python bench.py --no-cpu-baseline --no-other-configs --steps 21 --warmup 5 > $OUT/r5f_bench_mv3ddet_img0.json 2> $OUT/r5f_err0.txt; echo "rc $?"
ES_IMG_WGRAD=1 timeout 300 python bench.py --no-cpu-baseline --no-other-configs --steps 21 --warmup 5 > $OUT/r5f_bench_mv3ddet_img1.json 2> $OUT/r5f_err1.txt; echo "rc $?"
ES_BENCH_DIAG=1 timeout 500 python bench.py --no-cpu-baseline --only grounding --steps 40 --other-steps 40 --warmup 3 > $OUT/r5f_bench_grounding_diag.json 2> $OUT/r5f_bench_grounding_diag.err; echo "rc $?"
timeout 300 python bench.py --no-cpu-baseline --only from_files --steps 24 --other-steps 24 > $OUT/r5f_bench_from_files.json 2> $OUT/r5f_bench_from_files.err; echo "rc $?"
timeout 900 python -m pytest tests/test_gpu_insitu.py tests/test_gpu_config2.py tests/test_gpu_dataset.py -x -q > $OUT/r5f_tests_b.txt 2>&1; echo "rc $?"
tail -4 $OUT/r5f_tests_a.txt; tail -4 $OUT/r5f_tests_b.txt
