#!/bin/bash
# round 5, session t: the one slow step of the occupancy runs under a kernel trace (gap or slow kernel?)
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
db () { find /tmp/prof_$1 -name '*.db' | head -1; }
timeout 300 python -m pytest tests/test_gpu_optim_table.py -x -q > $OUT/r5t_tests.txt 2>&1; echo "rc $?"; tail -3 $OUT/r5t_tests.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_occ -o p -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --only occupancy --steps 40 --other-steps 40 --warmup 5 > $OUT/r5t_bench_occ.json 2> /tmp/prof_occ.log); echo "rc $?"
python tools/rocpd_outlier.py "$(db occ)" > $OUT/r5t_occ_outlier.txt 2>&1; cat $OUT/r5t_occ_outlier.txt | cut -c1-260
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r5t_bench_occ.json') if l.startswith('{')][-1])
print(d['ms_per_step'], d['step_ms'])
PY
