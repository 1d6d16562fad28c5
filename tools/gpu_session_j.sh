#!/bin/bash
# late round-3 experiment session: parity of the layout changes / LDS-DMA kernel / fused norm shadows / one-rank RCCL path, then A/B
# timings of the new options in one process per config
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_dma.py tests/test_gpu_ops.py tests/test_gpu_resnet2d.py tests/test_gpu_insitu.py \
    tests/test_gpu_zz_rccl.py -m gpu -q -s -p no:cacheprovider > gpurun_out/r3j_tests.txt 2>&1
echo "pytest rc $?" >> gpurun_out/r3j_tests.txt
grep -v "^$" gpurun_out/r3j_tests.txt | tail -25
timeout 200 python tools/sweep_options.py --steps 10 --warmup 3 --variants "10=1;10=2;NORM_SHADOW=0;10=2,NORM_SHADOW=0" \
    > gpurun_out/r3j_sweep_mv3ddet.txt 2> gpurun_out/r3j_sweep_mv3ddet.err
echo "sweep rc $?"; cat gpurun_out/r3j_sweep_mv3ddet.txt; tail -3 gpurun_out/r3j_sweep_mv3ddet.err
timeout 240 python tools/sweep_options.py --config occupancy --steps 6 --warmup 2 --variants "10=1;10=2" \
    > gpurun_out/r3j_sweep_occupancy.txt 2> gpurun_out/r3j_sweep_occupancy.err
echo "sweep rc $?"; cat gpurun_out/r3j_sweep_occupancy.txt; tail -3 gpurun_out/r3j_sweep_occupancy.err
