#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
B="python -X faulthandler bench.py --only grounding --no-cpu-baseline --steps 3 --warmup 1"
run () { env $1 timeout 200 $B > $OUT/r4p_$2.json 2> $OUT/r4p_$2.err; echo "== $2 ($1) rc $?"; grep -E "fault|Error|error" $OUT/r4p_$2.err | head -3; python -c "
import json
try:
    d=json.load(open('gpurun_out/r4p_$2.json')); print('   ms', d['ms_per_step'], d['step_ms'])
except Exception as e: print('   no json')"; }
run "A=1" default
run "A=1" default2
run "ES_WGRAD_TR=0" tr0
run "ES_NORM_CB_ROWS=0" cb0
run "ES_GEN_FUSED=0" gen0
run "ES_WGRAD_ASYNC=0" nowgstream
run "ES_TWO_STREAMS=0" noside
run "ES_GRAPHS=0" nograph
