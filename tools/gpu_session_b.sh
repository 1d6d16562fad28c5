#!/bin/bash
# round-3 session B: the tests that failed in session A (+ the new gather unit test), then the default bench line
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_fusion_losses.py tests/test_gpu_ops.py tests/test_gpu_config2.py tests/test_gpu_model.py tests/test_gpu_grounding.py tests/test_gpu_occ.py tests/test_gpu_config4.py tests/test_gpu_config5.py tests/test_gpu_dataset.py -m gpu -q -s -p no:cacheprovider > gpurun_out/r3_b_pytest.txt 2>&1
echo "pytest rc $?" >> gpurun_out/r3_b_pytest.txt
tail -12 gpurun_out/r3_b_pytest.txt
timeout 900 python bench.py > gpurun_out/r3_b_bench.json 2> gpurun_out/r3_b_bench.err
echo "bench rc $?"
tail -c 800 gpurun_out/r3_b_bench.err
head -c 400 gpurun_out/r3_b_bench.json
