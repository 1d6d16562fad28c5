#!/bin/bash
# round 5, session za: the step's main stream at a higher priority than the side / weight-gradient streams, with 4 and 8 hardware queues
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
python -c "import torch; print(torch.cuda.Stream(priority=-1).priority, torch.cuda.Stream(priority=-5).priority, torch.cuda.Stream(priority=3).priority)"
for q in 4 8; do for pr in -1 -2; do
  GPU_MAX_HW_QUEUES=$q ES_MAIN_PRIORITY=$pr timeout 300 python bench.py --no-cpu-baseline --no-other-configs --steps 30 --warmup 5 > $OUT/r5za_mv3ddet_q${q}_p$pr.json 2> $OUT/r5za_err.txt; echo "rc $?"
  GPU_MAX_HW_QUEUES=$q ES_MAIN_PRIORITY=$pr timeout 300 python bench.py --no-cpu-baseline --only grounding --steps 16 --other-steps 16 --warmup 4 > $OUT/r5za_grounding_q${q}_p$pr.json 2>> $OUT/r5za_err.txt; echo "rc $?"
done; done
tail -5 $OUT/r5za_err.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5za_*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
    except Exception as e:
        print(f, 'unreadable', e); continue
    s = sorted(d['step_ms'])
    print(f, d['value'], d['ms_per_step'], 'median', s[len(s) // 2], 'max', s[-1], 'min', s[0], d['losses'])
PY
