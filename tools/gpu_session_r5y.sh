#!/bin/bash
# round 5, session y: runtime knobs of the HIP queue layer -- more hardware queues than the default 4 (the step uses main + side + two
# weight-gradient streams + the copy stream), kernel arguments in device memory
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
run () {  # tag, env assignment
  env $2 timeout 300 python bench.py --no-cpu-baseline --no-other-configs --steps 30 --warmup 5 > $OUT/r5y_mv3ddet_$1.json 2> /dev/null; echo "rc $?"
  env $2 timeout 300 python bench.py --no-cpu-baseline --only grounding --steps 16 --other-steps 16 --warmup 4 > $OUT/r5y_grounding_$1.json 2> /dev/null; echo "rc $?"
  env $2 timeout 300 python bench.py --no-cpu-baseline --only occupancy --steps 20 --other-steps 20 --warmup 5 > $OUT/r5y_occ_$1.json 2> /dev/null; echo "rc $?"
}
run base ES_NOP=1
run q8 GPU_MAX_HW_QUEUES=8
run q2 GPU_MAX_HW_QUEUES=2
run kernarg HIP_FORCE_DEV_KERNARG=1
run base2 ES_NOP=2
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5y_*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
    except Exception as e:
        print(f, 'unreadable', e); continue
    s = sorted(d['step_ms'])
    print(f, d['value'], d['ms_per_step'], 'median', s[len(s) // 2], 'max', s[-1], 'min', s[0])
PY
