#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 800 python -m pytest tests/test_gpu_ops.py tests/test_gpu_resnet2d.py tests/test_gpu_model.py tests/test_gpu_config2.py -q -m gpu -x 2>&1 | tail -4
for v in 1 0 1; do
  ES_ROWGEMM=$v timeout 600 python bench.py --no-cpu-baseline --steps 9 --warmup 3 > gpurun_out/z_bench_r${v}.json 2> gpurun_out/z_bench.err
  python -c "
import json
d=json.loads(open('gpurun_out/z_bench_r${v}.json').read().strip().splitlines()[-1]); print('rowgemm=$v', d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['stage_ms']['A7 2-D backbone fwd'], d['stage_ms']['backward (head, 3-D, 2-D)'])"
done
tail -3 gpurun_out/z_bench.err
