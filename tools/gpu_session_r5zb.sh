#!/bin/bash
# round 5, session zb: the FPN's 3x3 output convolution on the dense engine (flat grid, row-sliced weight gradient): parity, A/B
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_dconv.py tests/test_gpu_occ.py tests/test_gpu_insitu.py -x -q > $OUT/r5zb_tests.txt 2>&1; echo "rc $?"; tail -4 $OUT/r5zb_tests.txt
for e in 1 0 1 0; do
  ES_FPN_DENSE=$e timeout 300 python bench.py --no-cpu-baseline --only occupancy --steps 20 --other-steps 20 --warmup 5 > $OUT/r5zb_occ_fpn${e}_$RANDOM.json 2> /dev/null; echo "rc $?"
done
ES_BENCH_DUMP=$OUT/r5zb_occ_launches.jsonl timeout 300 python bench.py --no-cpu-baseline --only occupancy --steps 8 --other-steps 8 --warmup 5 > $OUT/r5zb_occ_dump.json 2> /dev/null; echo "rc $?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5zb_occ_fpn*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
    except Exception as e:
        print(f, 'unreadable', e); continue
    s = sorted(d['step_ms'])
    print(f, d['value'], d['ms_per_step'], 'median', s[len(s) // 2], 'max', s[-1], d['stage_ms']['2-D backbone + FPN'], d['stage_ms']['backward'])
for l in open('gpurun_out/r5zb_occ_launches.jsonl'):
    r = json.loads(l)
    if r['K'] == 9 and r['cin'] == 256 and r['us'] > 100:
        print(r['fn'], r['n_out'], r['n_in'], r['us'])
PY
