#!/bin/bash
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_gpu_occ.py -m gpu -q -s > gpurun_out/c_pytest_occ.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c_pytest_occ.log
python -m pytest tests/test_gpu_config2.py tests/test_gpu_ops.py -m gpu -q -s -k "noise or large_maps" > gpurun_out/c_pytest_fix.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c_pytest_fix.log
timeout 900 python tools/bench_occ.py > gpurun_out/c_bench_occ.json 2> gpurun_out/c_bench_occ.err
grep -E "passed|failed|rc=" gpurun_out/c_pytest_occ.log gpurun_out/c_pytest_fix.log
tail -3 gpurun_out/c_bench_occ.err; cat gpurun_out/c_bench_occ.json
