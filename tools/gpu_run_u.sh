#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_gpu_dataset.py -q -m gpu 2>&1 | tail -4
for i in 1 2; do
  for v in -1 0; do
    ES_MAIN_PRIORITY=$v timeout 600 python bench.py --no-cpu-baseline --steps 9 --warmup 3 > gpurun_out/u_bench_p${v}_$i.json 2> gpurun_out/u_bench.err
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/u_bench_p*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d['step_ms'])
    except Exception as e: print(f, 'ERR', e)
PY
tail -3 gpurun_out/u_bench.err
