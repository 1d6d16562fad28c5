#!/bin/bash
# round 5, session q: the one slow step of every occupancy run (per-step allocator / collector state), the 1x1x1 down-sample after the
# slice-cap fix
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 300 python -m pytest tests/test_gpu_dconv.py -x -q > $OUT/r5q_tests.txt 2>&1; echo "rc $?"; tail -3 $OUT/r5q_tests.txt
ES_BENCH_DIAG=1 ES_BENCH_DUMP=$OUT/r5q_occ_launches.jsonl timeout 400 python bench.py --no-cpu-baseline --only occupancy --steps 24 --other-steps 24 --warmup 5 > $OUT/r5q_bench_occ_diag.json 2> $OUT/r5q_bench_occ.err; echo "rc $?"
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r5q_bench_occ_diag.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'])
for i, (s, g) in enumerate(zip(d['step_ms'], d['step_diag'])):
    print(i, s, {k: g[k] for k in ('host_ms', 'reserved_MB', 'allocated_MB', 'mallocs', 'frees', 'retries')}, g['graphs'], g['extra'])
print(d['gc_collections'])
PY
