#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "wgrad or bf16" 2>&1 | tail -3
timeout 600 python tools/bench_occ.py > gpurun_out/p_occ.json 2> gpurun_out/p_occ.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_p -o p -- python $R/bench.py --no-cpu-baseline --steps 12 --warmup 3 > $R/gpurun_out/p_bench.json 2> $R/gpurun_out/p_prof.err
cd $R
DB=$(find /tmp/prof_p -name '*.db' | head -1)
python tools/rocpd_stats.py $DB > gpurun_out/p_kernel_stats.txt 2>&1
python tools/rocpd_bygrid.py $DB > gpurun_out/p_bygrid.txt 2>&1
python -c "
import json
d=json.loads(open('gpurun_out/p_occ.json').read().strip().splitlines()[-1]); print('occ', d['ms_per_step'], d['stage_ms'], d['roofline']['achieved'])
d=json.loads(open('gpurun_out/p_bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms'])"
