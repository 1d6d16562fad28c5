#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 800 python -m pytest tests/test_gpu_ops.py tests/test_gpu_resnet2d.py tests/test_gpu_model.py -q -m gpu -x 2>&1 | tail -4
for i in 1 2; do
  for v in 1 0; do
    ES_ROWGEMM=$v timeout 600 python bench.py --no-cpu-baseline --steps 9 --warmup 3 > gpurun_out/y_bench_r${v}_$i.json 2> gpurun_out/y_bench.err
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/y_bench_r*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['stage_ms']['A7 2-D backbone fwd'], d['stage_ms']['backward (head, 3-D, 2-D)'])
    except Exception as e: print(f, 'ERR', e)
PY
tail -3 gpurun_out/y_bench.err
