#!/bin/bash
# closing session 1 of round 5: the whole GPU suite + smoke on the final tree
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/r5_gputest_full.txt 2>&1; echo "pytest rc $?"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/r5_smoke.txt 2>&1; echo "smoke rc $?"
grep -E "passed|failed|error" $OUT/r5_gputest_full.txt | tail -3
tail -4 $OUT/r5_smoke.txt
