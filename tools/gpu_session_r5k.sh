#!/bin/bash
# round 5, session k: why the from-files leg is slower behind the other legs of the default run than alone
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
ES_OTHER=from_files timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/r5k_a.json 2> $OUT/r5k_a.err; echo "rc $?"
ES_OTHER=from_files timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r5k_b.json 2> $OUT/r5k_b.err; echo "rc $?"
ES_OTHER=occupancy,from_files timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r5k_c.json 2> $OUT/r5k_c.err; echo "rc $?"
for f in a b c; do python - $OUT/r5k_$f.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
v = d['other_configs']['from_files']
print(sys.argv[1], d['value'], v.get('value'), v.get('ms_per_step'), v.get('vs_synthetic'), v.get('loader', {}).get('wait_ms_per_step'), v.get('error'))
PY
done
