#!/bin/bash
# round 5, first GPU minutes: parity of the dense-volume engine, its A/B against the map kernels, the occupancy step with it on / off,
# and the one-off A/B of the round-4 256-row gather tile (csrc/next)
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_dconv.py -x -q -s > $OUT/r5a_test_dconv.txt 2>&1; echo "rc $?"
timeout 600 python tools/bench_dconv.py --reps 5 > $OUT/r5a_dconv_ab.txt 2>&1; echo "rc $?"
ES_DENSE=0 timeout 500 python bench.py --no-cpu-baseline --only occupancy --steps 6 --warmup 3 > $OUT/r5a_bench_occ_dense0.json 2> $OUT/r5a_bench_occ_dense0.err; echo "rc $?"
ES_DENSE=1 timeout 500 python bench.py --no-cpu-baseline --only occupancy --steps 6 --warmup 3 > $OUT/r5a_bench_occ_dense1.json 2> $OUT/r5a_bench_occ_dense1.err; echo "rc $?"
timeout 900 python -m pytest tests/test_gpu_insitu.py tests/test_gpu_occ.py tests/test_gpu_config5.py -x -q > $OUT/r5a_tests_occ.txt 2>&1; echo "rc $?"
tail -5 $OUT/r5a_test_dconv.txt $OUT/r5a_tests_occ.txt
cat $OUT/r5a_dconv_ab.txt
