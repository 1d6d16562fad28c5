#!/bin/bash
# round 5, session s: AdamW + bf16 kernel copies in one pass (es_adamw_table): parity, then A/B against the two-pass path
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 400 python -m pytest tests/test_gpu_optim_table.py tests/test_gpu_fusion_losses.py -x -q > $OUT/r5s_tests.txt 2>&1; echo "rc $?"; tail -15 $OUT/r5s_tests.txt
for e in 1 0; do
  ES_ADAMW_CAST=$e timeout 300 python bench.py --no-cpu-baseline --only occupancy --steps 12 --other-steps 12 --warmup 5 > $OUT/r5s_occ_cast$e.json 2> /dev/null; echo "rc $?"
  ES_ADAMW_CAST=$e timeout 300 python bench.py --no-cpu-baseline --no-other-configs --steps 21 --warmup 5 > $OUT/r5s_mv3ddet_cast$e.json 2> /dev/null; echo "rc $?"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5s_*cast*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
    except Exception as e:
        print(f, 'unreadable', e); continue
    s = sorted(d['step_ms'])
    print(f, d['value'], d['ms_per_step'], 'median', s[len(s) // 2], d.get('optimizer_pass'), d['stage_ms'])
PY
