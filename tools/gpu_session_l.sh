#!/bin/bash
# A/B of the row-GEMM column-tile rule (option 12) and the DMA channel threshold (option 11), one process per config
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 200 python tools/sweep_options.py --steps 10 --warmup 3 --variants "12=64;12=256;12=1024;11=256" \
    > gpurun_out/r3l_sweep_mv3ddet.txt 2> gpurun_out/r3l_sweep_mv3ddet.err
echo "sweep rc $?"; cat gpurun_out/r3l_sweep_mv3ddet.txt; tail -3 gpurun_out/r3l_sweep_mv3ddet.err
timeout 240 python tools/sweep_options.py --config occupancy --steps 6 --warmup 2 --variants "12=1024;10=0" \
    > gpurun_out/r3l_sweep_occupancy.txt 2> gpurun_out/r3l_sweep_occupancy.err
echo "sweep rc $?"; cat gpurun_out/r3l_sweep_occupancy.txt; tail -3 gpurun_out/r3l_sweep_occupancy.err
