#!/bin/bash
# round-3 session H: in-situ launch check, the 2-rank bench path on one GPU over gloo (reducer groups, clip norm behind the
# collectives, exposed all-reduce time), a quick default-step timing
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_insitu.py -m gpu -q -s -p no:cacheprovider > gpurun_out/r3_h_pytest.txt 2>&1
echo "pytest rc $?" >> gpurun_out/r3_h_pytest.txt
tail -6 gpurun_out/r3_h_pytest.txt
ES_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 4 --warmup 2 > gpurun_out/r3_bench_2ranks_gloo_one_gpu.json 2> gpurun_out/r3_h_2ranks.err
echo "2-rank rc $?"; tail -c 600 gpurun_out/r3_h_2ranks.err; head -c 400 gpurun_out/r3_bench_2ranks_gloo_one_gpu.json
timeout 300 python bench.py --no-cpu-baseline --no-other-configs --steps 12 --warmup 3 > gpurun_out/r3_h_bench.json 2> gpurun_out/r3_h_bench.err
echo "bench rc $?"; head -c 300 gpurun_out/r3_h_bench.json
