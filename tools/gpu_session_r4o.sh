#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
HIP_LAUNCH_BLOCKING=1 timeout 300 python -X faulthandler bench.py --only grounding --no-cpu-baseline --steps 2 --warmup 1 > $OUT/r4o_ground.json 2> $OUT/r4o_ground.err; echo "rc $?"
tail -40 $OUT/r4o_ground.err
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_insitu.py -q -s -p no:cacheprovider -k "topk or insitu or specification" > $OUT/r4o_tests.txt 2>&1
echo "pytest rc $?" >> $OUT/r4o_tests.txt
grep -v Warning $OUT/r4o_tests.txt | grep -E "passed|failed|^E  |FAILED|^mv-|^occ" | head -30
