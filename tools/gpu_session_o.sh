#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_resnet2d.py tests/test_gpu_dma.py -m gpu -q -s -x -p no:cacheprovider -k "rowgemm or resnet50 or dma" > gpurun_out/r3o_tests.txt 2>&1
echo "pytest rc $?" >> gpurun_out/r3o_tests.txt
grep -v "^$" gpurun_out/r3o_tests.txt | tail -22
timeout 100 python tools/bench_rowgemm.py > gpurun_out/r3o_rowgemm_gen2.txt 2>&1; cat gpurun_out/r3o_rowgemm_gen2.txt
timeout 100 python tools/bench_rowgemm.py --opt 13=0 > gpurun_out/r3o_rowgemm_gen1.txt 2>&1; cat gpurun_out/r3o_rowgemm_gen1.txt
