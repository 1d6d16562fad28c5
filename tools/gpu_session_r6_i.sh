#!/bin/bash
# round 6, session i: narrow-input conv kernels (test + timing), weight-sharing order A/B, per-launch dump of the step
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_narrow.py tests/test_gpu_ops.py -m gpu -q -s -x > $OUT/r6i_narrow_test.txt 2>&1; echo "rc $?"; tail -5 $OUT/r6i_narrow_test.txt
timeout 300 python tools/bench_small_conv.py > $OUT/r6i_small_conv.txt 2>&1; cat $OUT/r6i_small_conv.txt
B="python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 5"
for rep in 1 2; do
  for v in "ES_NARROW=1 ES_WSHARE=0" "ES_NARROW=0 ES_WSHARE=0" "ES_NARROW=1 ES_WSHARE=1"; do
    env $v timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['value'])" | tee -a $OUT/r6i_ab.txt
  done
done
ES_BENCH_DUMP=$OUT/r6i_dump.jsonl timeout 300 $B > /dev/null 2>&1
python - <<'PY' | tee $OUT/r6i_slowest.txt
import json, os
rows = [json.loads(l) for l in open(os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'gpurun_out/r6i_dump.jsonl'))]
print(len(rows), 'engine launches, single-stream sum', round(sum(r['us'] for r in rows) / 1e3, 2), 'ms')
for r in sorted(rows, key=lambda r: -r['us'])[:70]:
    print(f"{r['us']:8.1f} us  {r['fn']:34s} K={r['K']:2d} {r['cin']:4d}->{r['cout']:4d} n_out={r['n_out']:7d} n_in={r['n_in']:7d} map={int(r['map'])}")
PY
