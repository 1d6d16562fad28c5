#!/bin/bash
# round 5, session c: the reference cycles a train step leaves behind (tools/gc_hunt.py), parity of the parity-class dense launches
# (strided data gradient, transposed convolution), occupancy step with them
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 300 python tools/gc_hunt.py grounding > $OUT/r5c_gc_grounding.txt 2>&1; echo "rc $?"
timeout 300 python tools/gc_hunt.py mv3ddet > $OUT/r5c_gc_mv3ddet.txt 2>&1; echo "rc $?"
timeout 900 python -m pytest tests/test_gpu_dconv.py -x -q -s > $OUT/r5c_test_dconv.txt 2>&1; echo "rc $?"
timeout 300 python tools/bench_dconv.py --reps 5 > $OUT/r5c_dconv_ab.txt 2>&1; echo "rc $?"
timeout 300 python bench.py --no-cpu-baseline --only occupancy --steps 8 --other-steps 8 --warmup 3 > $OUT/r5c_bench_occ.json 2> $OUT/r5c_bench_occ.err; echo "rc $?"
timeout 900 python -m pytest tests/test_gpu_insitu.py tests/test_gpu_occ.py -x -q -k "occupancy or config5 or neck or occ" > $OUT/r5c_tests_occ.txt 2>&1; echo "rc $?"
tail -3 $OUT/r5c_test_dconv.txt $OUT/r5c_tests_occ.txt
cat $OUT/r5c_gc_grounding.txt | tail -60
