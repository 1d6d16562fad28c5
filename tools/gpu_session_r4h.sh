#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
B="python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 5"
run () { env $1 timeout 200 $B > $OUT/r4h_$2.json 2> $OUT/r4h_$2.err; python -c "
import json; d=json.load(open('gpurun_out/r4h_$2.json')); print('$2', '$1', d['ms_per_step'], d['value'])"; }
run "ES_NEXT_PREFETCH=0" pf0
run "ES_NEXT_PREFETCH=1 ES_PF_GATE=1" pf1_gate
run "ES_NEXT_PREFETCH=1 ES_PF_GATE=1 ES_PF_PRIORITY=0" pf1_gate_prio0
run "ES_NEXT_PREFETCH=1 ES_PF_GATE=0" pf1_nogate
run "ES_NEXT_PREFETCH=0 ES_TWO_STREAMS=0" pf0_one_side
run "ES_NEXT_PREFETCH=0 ES_WGRAD_ASYNC=0" pf0_no_wgrad_streams
run "ES_NEXT_PREFETCH=0 GPU_MAX_HW_QUEUES=2" pf0_q2
run "ES_NEXT_PREFETCH=0 GPU_MAX_HW_QUEUES=3" pf0_q3
ES_PF_GATE=1 timeout 300 python tools/host_profile.py > $OUT/r4h_host_profile_gate.txt 2>&1; echo "rc $?"
sed -n 6,24p $OUT/r4h_host_profile_gate.txt
timeout 1500 python -m pytest tests/test_gpu_grounding.py tests/test_gpu_insitu.py -q -s -p no:cacheprovider > $OUT/r4h_tests.txt 2>&1
echo "pytest rc $?" >> $OUT/r4h_tests.txt
grep -v Warning $OUT/r4h_tests.txt | grep -E "^mv-|^occupancy|passed|failed|^E  |FAILED|bf16" | head -40
