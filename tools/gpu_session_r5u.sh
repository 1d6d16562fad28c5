#!/bin/bash
# round 5, session u: are the slow steps late wake-ups of the host thread at its first stream synchronisation (the device idles 7 - 22 ms
# there, profiles/r5t_occ_outlier.txt)?  ROCr waits on interrupts by default; HSA_ENABLE_INTERRUPT=0 makes the waits poll.
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
for v in 1 0; do
  HSA_ENABLE_INTERRUPT=$v timeout 300 python bench.py --no-cpu-baseline --only occupancy --steps 40 --other-steps 40 --warmup 5 > $OUT/r5u_occ_int$v.json 2> /dev/null; echo "rc $?"
  HSA_ENABLE_INTERRUPT=$v timeout 300 python bench.py --no-cpu-baseline --no-other-configs --steps 30 --warmup 5 > $OUT/r5u_mv3ddet_int$v.json 2> /dev/null; echo "rc $?"
  HSA_ENABLE_INTERRUPT=$v timeout 300 python bench.py --no-cpu-baseline --only grounding --steps 20 --other-steps 20 --warmup 4 > $OUT/r5u_grounding_int$v.json 2> /dev/null; echo "rc $?"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5u_*_int*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
    except Exception as e:
        print(f, 'unreadable', e); continue
    s = sorted(d['step_ms'])
    print(f, d['value'], d['ms_per_step'], 'median', s[len(s) // 2], 'max', s[-1], 'min', s[0])
    print('   ', d['step_ms'])
PY
