#!/bin/bash
# A/B on one box: weight gradients from bf16 shadows on / off
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for i in 1 2; do
  for v in 1 0; do
    ES_WGRAD_SHADOW=$v timeout 600 python bench.py --no-cpu-baseline --steps 8 --warmup 3 > gpurun_out/n_bench_${v}_$i.json 2> gpurun_out/n_bench.err
  done
done
for v in 1 0; do
  ES_WGRAD_SHADOW=$v timeout 600 python tools/bench_occ.py > gpurun_out/n_occ_$v.json 2> gpurun_out/n_occ.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/n_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['roofline']
        print(f, d['ms_per_step'], 'engine ms', r.get('kernel_ms_per_step', r.get('kernel_ms')), 'launches', r.get('launches_per_step', r.get('launches')), 'achieved', r['achieved'], r.get('frac_of_binding_roof'), d['stage_ms'].get('A10-A16 head fwd + targets + losses'))
    except Exception as e: print(f, 'ERR', e)
PY
