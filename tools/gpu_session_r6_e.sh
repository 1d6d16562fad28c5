#!/bin/bash
# round 6 session e: determinism bisect of the grounding step across detector builds + the whole GPU suite on the current tree
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 600 python tools/bisect_determinism.py mv_grounding.py --backward > $OUT/r6e_bisect_grounding.txt 2>&1; echo "bisect rc $?"
tail -40 $OUT/r6e_bisect_grounding.txt
timeout 1800 python -m pytest tests -m gpu -q -x > $OUT/r6e_gputest_full.txt 2>&1; echo "pytest rc $?"
tail -8 $OUT/r6e_gputest_full.txt
timeout 300 python tools/bench_halo.py 2>&1 | tail -7 | cut -c1-130
