#!/bin/bash
# round 5, session h: collector freeze (all detectors), 3x3 image weight gradients per kernel row (A/B + launch dumps), the occupancy
# detector with bf16 activation rows in its image backbone (A/B + its tests)
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
ES_BENCH_DIAG=1 timeout 500 python bench.py --no-cpu-baseline --only grounding --steps 40 --other-steps 40 --warmup 3 > $OUT/r5h_bench_grounding_diag.json 2> $OUT/r5h_bench_grounding_diag.err; echo "rc $?"
ES_IMG_WGRAD=0 ES_BENCH_DUMP=$OUT/r5h_launches_img0.jsonl timeout 300 python bench.py --no-cpu-baseline --no-other-configs --steps 21 --warmup 5 > $OUT/r5h_bench_mv3ddet_img0.json 2> $OUT/r5h_err0.txt; echo "rc $?"
ES_IMG_WGRAD=1 ES_BENCH_DUMP=$OUT/r5h_launches_img1.jsonl timeout 300 python bench.py --no-cpu-baseline --no-other-configs --steps 21 --warmup 5 > $OUT/r5h_bench_mv3ddet_img1.json 2> $OUT/r5h_err1.txt; echo "rc $?"
ES_OCC_ACT16=0 timeout 300 python bench.py --no-cpu-baseline --only occupancy --steps 10 --other-steps 10 --warmup 3 > $OUT/r5h_bench_occ_act0.json 2> $OUT/r5h_erro0.txt; echo "rc $?"
ES_OCC_ACT16=1 timeout 300 python bench.py --no-cpu-baseline --only occupancy --steps 10 --other-steps 10 --warmup 3 > $OUT/r5h_bench_occ_act1.json 2> $OUT/r5h_erro1.txt; echo "rc $?"
timeout 900 python -m pytest tests/test_gpu_occ.py tests/test_gpu_config5.py tests/test_gpu_insitu.py -x -q -k "occ or config5" > $OUT/r5h_tests_occ.txt 2>&1; echo "rc $?"
tail -5 $OUT/r5h_tests_occ.txt
