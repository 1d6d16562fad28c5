#!/bin/bash
# round 6, session t: how often does the grounding step fall into its slow mode (~ +9 ms on every step of a process)?
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
B="python bench.py --no-cpu-baseline --only grounding --steps 6 --warmup 3 --other-steps 6"
for rep in 1 2 3 4 5 6 7 8; do
  for v in "ES_TEXT_STREAM=wgrad" "ES_TEXT_STREAM=own" "ES_TEXT_STREAM=own GPU_MAX_HW_QUEUES=8"; do
    env $v timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'])" | tee -a $OUT/r6t_modes.txt
  done
done
sort $OUT/r6t_modes.txt | awk '{k=$1" "$2; if ($NF+0 > 50) s[k]++; n[k]++} END {for (k in n) print k, "runs", n[k], "slow", s[k]+0}'
