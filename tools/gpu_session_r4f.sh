#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 300 python tools/host_profile.py > $OUT/r4f_host_profile.txt 2>&1; echo "rc $?"
head -24 $OUT/r4f_host_profile.txt
timeout 1500 python -m pytest tests/test_gpu_grounding.py tests/test_gpu_insitu.py tests/test_gpu_configs.py -q -s -p no:cacheprovider > $OUT/r4f_tests.txt 2>&1
echo "pytest rc $?" >> $OUT/r4f_tests.txt
grep -v Warning $OUT/r4f_tests.txt | tail -60
