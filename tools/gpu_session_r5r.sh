#!/bin/bash
# round 5, session r: the torch operators of one step by call site (copies / fills beside the es_* launches)
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 300 python tools/copy_hunt.py mv3ddet > $OUT/r5r_copy_hunt_mv3ddet.txt 2> $OUT/r5r_copy_hunt_mv3ddet.err; echo "rc $?"
timeout 300 python tools/copy_hunt.py grounding > $OUT/r5r_copy_hunt_grounding.txt 2> $OUT/r5r_copy_hunt_grounding.err; echo "rc $?"
head -60 $OUT/r5r_copy_hunt_mv3ddet.txt | cut -c1-200; tail -5 $OUT/r5r_copy_hunt_mv3ddet.err
