#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 300 python tools/host_profile.py > $OUT/r4c_host_profile.txt 2>&1; echo "rc $?"
head -60 $OUT/r4c_host_profile.txt
timeout 300 python -m pytest tests/test_gpu_grounding.py -q -s -x -p no:cacheprovider -k "layernorm or train_step" > $OUT/r4c_tests.txt 2>&1
echo "pytest rc $?" >> $OUT/r4c_tests.txt
tail -12 $OUT/r4c_tests.txt
