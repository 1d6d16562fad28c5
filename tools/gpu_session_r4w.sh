#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_insitu.py -q -s -x -p no:cacheprovider -k "norm or parity or specification or topk" > $OUT/r4w_tests.txt 2>&1
echo "pytest rc $?" >> $OUT/r4w_tests.txt
grep -v Warning $OUT/r4w_tests.txt | grep -E "passed|failed|^E  |FAILED" | head
timeout 300 python tools/sweep_options.py --steps 12 --warmup 3 --variants "15=0;15=8192;15=16384;15=8192,17=0" > $OUT/r4w_sweep.txt 2> $OUT/r4w_sweep.err
cat $OUT/r4w_sweep.txt; tail -2 $OUT/r4w_sweep.err
