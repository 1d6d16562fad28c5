#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_o -o p -- python $R/tools/bench_occ.py > $R/gpurun_out/o_occ.json 2> $R/gpurun_out/o_prof.err
cd $R
DB=$(find /tmp/prof_o -name '*.db' | head -1)
python tools/rocpd_stats.py $DB > gpurun_out/o_occ_kernel_stats.txt 2>&1
python tools/rocpd_bygrid.py $DB > gpurun_out/o_occ_wgrad_bygrid.txt 2>&1
timeout 600 python bench.py --no-cpu-baseline --steps 12 --warmup 3 > gpurun_out/o_bench.json 2> gpurun_out/o_bench.err
head -30 gpurun_out/o_occ_kernel_stats.txt | cut -c1-160
python -c "
import json
d=json.loads(open('gpurun_out/o_bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms'])"
