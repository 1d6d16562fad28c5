#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 150 python -m pytest tests/test_gpu_ops.py tests/test_gpu_config2.py -m gpu -q -s -x -p no:cacheprovider -k "strided_chain or voxelize or config2 or batch" > gpurun_out/r3w_tests.txt 2>&1
echo "pytest rc $?" >> gpurun_out/r3w_tests.txt
grep -v "^$" gpurun_out/r3w_tests.txt | tail -8
for cb in 1 0; do
ES_COORD_BATCH=$cb timeout 100 python bench.py --no-cpu-baseline --no-other-configs --steps 10 --warmup 3 > gpurun_out/r3w_bench_cb$cb.json 2> /dev/null; echo rc $?
python -c "import json;d=json.load(open('gpurun_out/r3w_bench_cb$cb.json'));print('coord batch $cb', d['value'], d['ms_per_step'], d['step_ms'], d['parity']['ok'])"
done
