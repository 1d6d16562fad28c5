#!/bin/bash
# round-3 session F: the in-situ launch check + the tests whose bounds changed, then the tuning sweeps
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_insitu.py tests/test_gpu_resnet2d.py tests/test_gpu_model.py tests/test_gpu_occ.py -m gpu -q -s -p no:cacheprovider > gpurun_out/r3_f_pytest.txt 2>&1
echo "pytest rc $?" >> gpurun_out/r3_f_pytest.txt
tail -8 gpurun_out/r3_f_pytest.txt
timeout 600 python tools/sweep_options.py --steps 12 > gpurun_out/r3_f_sweep_mv3ddet.txt 2> gpurun_out/r3_f_sweep_mv3ddet.err
echo "sweep rc $?"; cat gpurun_out/r3_f_sweep_mv3ddet.txt
timeout 600 python tools/sweep_options.py --config occupancy --steps 8 > gpurun_out/r3_f_sweep_occ.txt 2> gpurun_out/r3_f_sweep_occ.err
echo "sweep occ rc $?"; cat gpurun_out/r3_f_sweep_occ.txt
