#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_config2.py tests/test_gpu_insitu.py -q -s -x -p no:cacheprovider -k "not config4_scale and not config5_scale" > $OUT/r4j_tests.txt 2>&1
echo "pytest rc $?" >> $OUT/r4j_tests.txt
grep -v Warning $OUT/r4j_tests.txt | grep -E "passed|failed|^E  |FAILED|norm" | head -30
timeout 300 python tools/sweep_options.py --steps 12 --warmup 3 --variants "15=0;15=4096;15=65536" > $OUT/r4j_sweep.txt 2> $OUT/r4j_sweep.err
cat $OUT/r4j_sweep.txt; tail -3 $OUT/r4j_sweep.err
timeout 300 python -m pytest tests/test_gpu_grounding.py -q -s -p no:cacheprovider -k "train_step" > $OUT/r4j_tests2.txt 2>&1
echo "pytest rc $?" >> $OUT/r4j_tests2.txt
grep -v Warning $OUT/r4j_tests2.txt | grep -E "passed|failed|^E  |FAILED|bf16" | head -20
