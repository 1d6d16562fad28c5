#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "wgrad" 2>&1 | tail -4
for v in 1 0 1 0; do
  ES_WGRAD_HUGE=$v timeout 600 python tools/bench_occ.py 2> gpurun_out/v_occ.err | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('huge=$v', d['ms_per_step'], d['stage_ms']['backward'], d['roofline']['achieved'], d['roofline'].get('kernel_ms'))"
done
tail -2 gpurun_out/v_occ.err
