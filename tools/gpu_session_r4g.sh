#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
for pr in -1 0; do
ES_PF_PRIORITY=$pr timeout 300 python tools/host_profile.py > $OUT/r4g_host_profile_prio${pr}.txt 2>&1; echo "rc $?"
sed -n 6,24p $OUT/r4g_host_profile_prio${pr}.txt
done
B="python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 5"
ES_PF_PRIORITY=0 timeout 200 $B > $OUT/r4g_bench_prio0.json 2> $OUT/r4g_bench_prio0.err; echo "rc $?"
python -c "
import json; d=json.load(open('gpurun_out/r4g_bench_prio0.json')); print('prio0 pf1', d['ms_per_step'], d['value'])"
