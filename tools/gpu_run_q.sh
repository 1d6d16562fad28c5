#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_fusion_losses.py tests/test_gpu_grounding.py tests/test_gpu_model.py -q -m gpu 2>&1 | tail -8
cd /tmp
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_q -o p -- python $R/bench.py --no-cpu-baseline --steps 6 --warmup 3 > $R/gpurun_out/q_bench.json 2> $R/gpurun_out/q_prof.err
cd $R
DB=$(find /tmp/prof_q -name '*.db' | head -1)
python tools/rocpd_stats.py $DB 2>&1 | grep -i "k_pos_losses\|k_focal\|k_box_cd"
timeout 600 python bench.py --no-cpu-baseline --steps 9 --warmup 3 > gpurun_out/q_bench2.json 2> gpurun_out/q_bench2.err
python -c "
import json
d=json.loads(open('gpurun_out/q_bench2.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms'], d['losses'])"
