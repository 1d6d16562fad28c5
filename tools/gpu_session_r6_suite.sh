#!/bin/bash
# round 6: the whole GPU suite + smoke on the current tree
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
T=${TAG:-r6}
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 1800 python -m pytest tests -m gpu -q -s > $OUT/${T}_gputest_full.txt 2>&1; echo "pytest rc $?"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${T}_smoke.txt 2>&1; echo "smoke rc $?"
grep -E "passed|failed|error" $OUT/${T}_gputest_full.txt | tail -3
tail -4 $OUT/${T}_smoke.txt
