#!/bin/bash
# round 5, session z: GPU_MAX_HW_QUEUES 3 / 5 / 6 (4 is the default; 2 and 8 lose: profiles/r5y_*)
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
for q in 3 5 6 4; do
  GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --no-cpu-baseline --no-other-configs --steps 30 --warmup 5 > $OUT/r5z_mv3ddet_q$q.json 2> /dev/null; echo "rc $?"
  GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --no-cpu-baseline --only grounding --steps 16 --other-steps 16 --warmup 4 > $OUT/r5z_grounding_q$q.json 2> /dev/null; echo "rc $?"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5z_*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
    except Exception as e:
        print(f, 'unreadable', e); continue
    s = sorted(d['step_ms'])
    print(f, d['value'], d['ms_per_step'], 'median', s[len(s) // 2], 'max', s[-1], 'min', s[0])
PY
