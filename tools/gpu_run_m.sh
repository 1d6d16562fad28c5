#!/bin/bash
# wgrad from bf16 shadows: parity + benches (default, shadow wgrad off via ES_SHADOW=0 for reference, occupancy)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_config2.py -q -m gpu -x 2>&1 | tail -5 > gpurun_out/m_tests.txt
timeout 600 python bench.py --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/m_bench.json 2> gpurun_out/m_bench.err
timeout 600 python tools/bench_occ.py > gpurun_out/m_occ.json 2> gpurun_out/m_occ.err
cat gpurun_out/m_tests.txt; python - <<'PY'
import json
for f in ('m_bench','m_occ'):
    try:
        d=json.loads(open(f'gpurun_out/{f}.json').read().strip().splitlines()[-1])
        print(f, d['ms_per_step'], d.get('stage_ms'), d['roofline'].get('achieved'))
    except Exception as e: print(f, 'ERR', e)
PY
