#!/bin/bash
# round 5, session n: does the stream -> hardware-queue assignment (creation order) move the mv-3ddet / grounding step?
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 300 python -m pytest tests/test_gpu_dconv.py -x -q > $OUT/r5n_test_dconv.txt 2>&1; echo "rc $?"
for k in 0 1 2 3 5; do
  ES_STREAM_SKEW=$k timeout 200 python bench.py --no-cpu-baseline --no-other-configs --steps 21 --warmup 5 > $OUT/r5n_mv3ddet_skew$k.json 2> /dev/null; echo "rc $?"
done
for k in 0 1 2 3; do
  ES_STREAM_SKEW=$k timeout 300 python bench.py --no-cpu-baseline --only grounding --steps 12 --other-steps 12 --warmup 3 > $OUT/r5n_grounding_skew$k.json 2> /dev/null; echo "rc $?"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5n_*skew*.json')):
    d = json.loads([l for l in open(f) if l.startswith('{')][-1])
    s = sorted(d['step_ms'])
    print(f, d['value'], d['ms_per_step'], 'median', s[len(s) // 2])
PY
tail -2 $OUT/r5n_test_dconv.txt
