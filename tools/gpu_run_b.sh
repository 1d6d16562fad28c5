#!/bin/bash
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s > gpurun_out/b_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/b_pytest.log
ES_SHADOW=1 timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_config2.py -m gpu -q -s > gpurun_out/b_pytest_shadow.log 2>&1; echo "pytest rc=$?" >> gpurun_out/b_pytest_shadow.log
grep -E "passed|failed|rc=" gpurun_out/b_pytest.log gpurun_out/b_pytest_shadow.log
