#!/bin/bash
# round 5, session p: the frozen text encoder queued under the image backbone (grounding A/B), the neck's 1x1x1 stride-2 down-sample
# on the dense engine (parity + occupancy step), the neck family counted on the neck's own launches
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 300 python -m pytest tests/test_gpu_dconv.py tests/test_gpu_grounding.py -x -q > $OUT/r5p_tests.txt 2>&1; echo "rc $?"; tail -3 $OUT/r5p_tests.txt
for e in 1 0 1 0; do
  ES_TEXT_EARLY=$e timeout 300 python bench.py --no-cpu-baseline --only grounding --steps 12 --other-steps 12 --warmup 3 > $OUT/r5p_grounding_early${e}_$RANDOM.json 2> /dev/null; echo "rc $?"
done
ES_BENCH_DUMP=$OUT/r5p_occ_launches.jsonl timeout 400 python bench.py --no-cpu-baseline --only occupancy --steps 10 --other-steps 10 --warmup 4 > $OUT/r5p_bench_occ.json 2> $OUT/r5p_bench_occ.err; echo "rc $?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5p_grounding_early*.json')) + ['gpurun_out/r5p_bench_occ.json']:
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
    except Exception as e:
        print(f, 'unreadable', e); continue
    s = sorted(d['step_ms'])
    print(f, d['value'], d['ms_per_step'], 'median', s[len(s) // 2], d['roofline'].get('frac'), d['roofline'].get('kernel_ms_per_step'))
PY
