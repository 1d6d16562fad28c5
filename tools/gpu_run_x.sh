#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_resnet2d.py -q -m gpu -k "rowgemm or resnet or bf16_mode" 2>&1 | tail -6
for i in 1 2; do
  for v in 1 0; do
    ES_ROWGEMM=$v timeout 600 python bench.py --no-cpu-baseline --steps 9 --warmup 3 > gpurun_out/x_bench_r${v}_$i.json 2> gpurun_out/x_bench.err
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/x_bench_r*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['stage_ms']['A7 2-D backbone fwd'], d['losses'])
    except Exception as e: print(f, 'ERR', e)
PY
tail -3 gpurun_out/x_bench.err
