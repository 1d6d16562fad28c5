#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_dataset.py tests/test_gpu_fusion_losses.py -q -m gpu 2>&1 | tail -12
timeout 900 python tools/bench_loader.py --scans 6 --frames 22 --threads 1,8,32,64,128 > gpurun_out/r_loader.json 2> gpurun_out/r_loader.err
tail -8 gpurun_out/r_loader.err; cat gpurun_out/r_loader.json | head -c 1500; echo
cd /tmp
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_r -o p -- python $R/bench.py --no-cpu-baseline --steps 6 --warmup 3 > $R/gpurun_out/r_bench.json 2> $R/gpurun_out/r_prof.err
cd $R
DB=$(find /tmp/prof_r -name '*.db' | head -1)
python tools/rocpd_stats.py $DB 2>&1 | grep -i "k_pos_losses\|k_pos_compact\|k_focal"
python -c "
import json
d=json.loads(open('gpurun_out/r_bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms'])"
