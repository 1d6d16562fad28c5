#!/bin/bash
# round 6, session m: where the grounding step's torch copies / fills come from; its slowest engine launches
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 300 python tools/copy_hunt.py grounding 12 > $OUT/r6m_copy_hunt_grounding.txt 2>&1; echo "rc $?"; tail -60 $OUT/r6m_copy_hunt_grounding.txt | cut -c1-230
ES_BENCH_DUMP=$OUT/r6m_dump_grounding.jsonl timeout 300 python bench.py --no-cpu-baseline --only grounding --steps 5 --warmup 2 --other-steps 5 > /dev/null 2>&1
python - <<'PY' | tee $OUT/r6m_slowest_grounding.txt
import json, os, collections
rows = [json.loads(l) for l in open(os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'gpurun_out/r6m_dump_grounding.jsonl'))]
print(len(rows), 'engine launches, single-stream sum', round(sum(r['us'] for r in rows) / 1e3, 2), 'ms')
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    k = (r['fn'], r['K'], r['cin'], r['cout'], r['n_out'], r['n_in'], int(r['map']))
    agg[k][0] += 1; agg[k][1] += r['us']
for k, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{us:9.1f} us  x{c:3d}  {k[0]:30s} K={k[1]:2d} {k[2]:4d}->{k[3]:4d} n_out={k[4]:7d} n_in={k[5]:7d} map={k[6]}")
PY
