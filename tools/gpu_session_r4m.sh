#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 600 python tools/sweep_options.py --steps 12 --warmup 3 --variants "4=2048;4=1024;4=2048,6=1024;6=1024;6=512;4=1024,6=512;7=64;8=768;8=192;4=2048,5=1024" > $OUT/r4m_sweep.txt 2> $OUT/r4m_sweep.err
cat $OUT/r4m_sweep.txt; tail -3 $OUT/r4m_sweep.err
timeout 300 python -m pytest tests/test_gpu_dma.py tests/test_gpu_grounding.py -q -s -x -p no:cacheprovider > $OUT/r4m_tests.txt 2>&1
echo "pytest rc $?" >> $OUT/r4m_tests.txt
grep -v Warning $OUT/r4m_tests.txt | grep -E "passed|failed|^E  |FAILED|teacher" | head -30
