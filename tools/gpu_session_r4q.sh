#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
run () { env $1 timeout 300 python bench.py --only $3 --no-cpu-baseline --steps 4 --warmup 1 > $OUT/r4q_$2.json 2> $OUT/r4q_$2.err; echo "== $2 ($1) rc $?"; grep -E "fault|Error|error" $OUT/r4q_$2.err | head -3; python -c "
import json
try:
    d=json.load(open('gpurun_out/r4q_$2.json')); print('   ms', d['ms_per_step'], d['step_ms'], d['engine_all']['launches_per_step'])
except Exception as e: print('   no json')"; }
run "A=1" g_default grounding
run "ES_WG_BIG_TARGET=8192 ES_WG_SMALL_TARGET=4096" g_oldslices grounding
run "ES_WG_BIG_TARGET=8192" g_big8192 grounding
run "ES_WG_SMALL_TARGET=4096" g_small4096 grounding
run "A=1" o_default occupancy
run "ES_WG_BIG_TARGET=8192 ES_WG_SMALL_TARGET=4096" o_oldslices occupancy
