#!/bin/bash
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s > gpurun_out/h_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/h_pytest.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/h_bench.json 2> gpurun_out/h_bench.err
ES_DET_SPLIT=0 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/h_bench_nodet.json 2>> gpurun_out/h_bench.err
timeout 900 python tools/bench_grounding.py > gpurun_out/h_bench_ground.json 2> gpurun_out/h_bench_ground.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_h -o p -- python $R/bench.py --no-cpu-baseline --steps 4 --warmup 2 > $R/gpurun_out/h_prof.json 2> $R/gpurun_out/h_prof.err
cd $R
DB=$(find /tmp/prof_h -name '*.db' | head -1); python tools/rocpd_stats.py $DB gpurun_out/h_kernel_stats.txt > /dev/null 2>&1
grep -E "passed|failed|rc=" gpurun_out/h_pytest.log; grep -E "run-to-run" gpurun_out/h_pytest.log
for f in h_bench h_bench_nodet; do python -c "
import json
d=json.loads(open('gpurun_out/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac_of_binding_roof'], d['stage_ms'])"; done
python -c "
import json
d=json.loads(open('gpurun_out/h_bench_ground.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['hungarian_ms'], d['roofline']['kernel_ms'])"
