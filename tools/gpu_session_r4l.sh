#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 300 python tools/sweep_options.py --steps 12 --warmup 3 --variants "16=0;15=0;16=0,15=0" > $OUT/r4l_sweep.txt 2> $OUT/r4l_sweep.err
cat $OUT/r4l_sweep.txt; tail -3 $OUT/r4l_sweep.err
timeout 900 python -m pytest tests/test_gpu_dma.py tests/test_gpu_experimental.py tests/test_gpu_model.py tests/test_gpu_config2.py tests/test_gpu_fusion_losses.py tests/test_gpu_occ.py -q -s -x -p no:cacheprovider > $OUT/r4l_tests.txt 2>&1
echo "pytest rc $?" >> $OUT/r4l_tests.txt
grep -v Warning $OUT/r4l_tests.txt | grep -E "passed|failed|^E  |FAILED" | head -30
