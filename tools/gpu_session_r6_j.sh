#!/bin/bash
# round 6, session j: narrow kernels v2, 320-column tiles (forward + weight gradient): tests, A/B tools, step A/B
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_narrow.py tests/test_gpu_imgwgrad.py tests/test_gpu_ops.py tests/test_gpu_config2.py -m gpu -q -s -x > $OUT/r6j_tests.txt 2>&1; echo "rc $?"; tail -3 $OUT/r6j_tests.txt; grep "scans:" $OUT/r6j_tests.txt
timeout 300 python tools/bench_imgwgrad.py 2>&1 | tail -18 > $OUT/r6j_rows_ab.txt; cat $OUT/r6j_rows_ab.txt
B="python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 5"
for rep in 1 2 3; do
  for v in "ES_RG320=1" "ES_RG320=0"; do
    env $v timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['value'])" | tee -a $OUT/r6j_ab.txt
  done
done
ES_BENCH_DUMP=$OUT/r6j_dump.jsonl timeout 300 $B > /dev/null 2>&1
python - <<'PY' | tee $OUT/r6j_slowest.txt
import json, os
rows = [json.loads(l) for l in open(os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'gpurun_out/r6j_dump.jsonl'))]
print(len(rows), 'engine launches, single-stream sum', round(sum(r['us'] for r in rows) / 1e3, 2), 'ms')
for r in sorted(rows, key=lambda r: -r['us'])[:40]:
    print(f"{r['us']:8.1f} us  {r['fn']:34s} K={r['K']:2d} {r['cin']:4d}->{r['cout']:4d} n_out={r['n_out']:7d} n_in={r['n_in']:7d} map={int(r['map'])}")
PY
