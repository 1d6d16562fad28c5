#!/bin/bash
# round 6, session u: the grounding leg inside the driver's default run (CPU baseline leg included): slow mode or not?
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
P="import sys,json; d=json.loads(sys.stdin.read()); o=d.get('other_configs',{}); print(TAG, 'mv3ddet', d['ms_per_step'], {k: v.get('ms_per_step') for k, v in o.items()})"
for rep in 1 2 3; do
  timeout 600 python bench.py 2>/dev/null | tail -1 | python -c "TAG='default'; $P" | tee -a $OUT/r6u_modes.txt
done
for rep in 1 2; do
  ES_OTHER=grounding timeout 600 python bench.py 2>/dev/null | tail -1 | python -c "TAG='baseline + grounding only'; $P" | tee -a $OUT/r6u_modes.txt
  ES_OTHER=grounding timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "TAG='no baseline, grounding only'; $P" | tee -a $OUT/r6u_modes.txt
  ES_TEXT_ASYNC=0 timeout 600 python bench.py 2>/dev/null | tail -1 | python -c "TAG='default, ES_TEXT_ASYNC=0'; $P" | tee -a $OUT/r6u_modes.txt
done
