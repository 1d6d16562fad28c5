#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_k -o p -- python $R/bench.py --no-cpu-baseline --steps 4 --warmup 2 --resident > /dev/null 2> $R/gpurun_out/k_prof.err
cd $R
DB=$(find /tmp/prof_k -name '*.db' | head -1)
python tools/rocpd_timeline.py $DB 5 > gpurun_out/k_timeline.txt 2>&1
python tools/rocpd_busy.py $DB 5 > gpurun_out/k_busy.txt 2>&1
head -50 gpurun_out/k_timeline.txt; head -12 gpurun_out/k_busy.txt
