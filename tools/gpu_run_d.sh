#!/bin/bash
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_gpu_grounding.py -m gpu -q -s > gpurun_out/d_pytest_ground.log 2>&1; echo "pytest rc=$?" >> gpurun_out/d_pytest_ground.log
python -m pytest tests/test_gpu_config2.py tests/test_gpu_occ.py -m gpu -q -s -k "noise or occ_detector" > gpurun_out/d_pytest_fix.log 2>&1; echo "pytest rc=$?" >> gpurun_out/d_pytest_fix.log
grep -E "passed|failed|rc=" gpurun_out/d_pytest_ground.log gpurun_out/d_pytest_fix.log
