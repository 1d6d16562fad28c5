#!/bin/bash
# round 5, session x: the host thread pool of the train steps -- torch's default (ES_HOST_THREADS=0: 128 on these boxes), the engine's
# cap (auto: granted cores / 4) and one thread -- step time and CFS throttling of the three configurations
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
for v in 0 auto 1; do
  ES_HOST_THREADS=$v timeout 300 python bench.py --no-cpu-baseline --only occupancy --steps 40 --other-steps 40 --warmup 5 > $OUT/r5x_occ_threads$v.json 2> /dev/null; echo "rc $?"
  ES_HOST_THREADS=$v timeout 300 python bench.py --no-cpu-baseline --no-other-configs --steps 30 --warmup 5 > $OUT/r5x_mv3ddet_threads$v.json 2> /dev/null; echo "rc $?"
  ES_HOST_THREADS=$v timeout 300 python bench.py --no-cpu-baseline --only grounding --steps 20 --other-steps 20 --warmup 4 > $OUT/r5x_grounding_threads$v.json 2> /dev/null; echo "rc $?"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5x_*_threads*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
    except Exception as e:
        print(f, 'unreadable', e); continue
    s = sorted(d['step_ms'])
    print(f, d['value'], d['ms_per_step'], 'median', s[len(s) // 2], 'max', s[-1], 'min', s[0], d.get('host'))
PY
