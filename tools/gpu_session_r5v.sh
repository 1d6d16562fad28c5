#!/bin/bash
# round 5, session v: where the host thread loses the time of the slow occupancy steps
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 300 python tools/host_stalls.py occupancy 60 2.0 > $OUT/r5v_host_stalls_occ.txt 2> $OUT/r5v_host_stalls_occ.err; echo "rc $?"
HS_GC_LOG=0 timeout 300 python tools/host_stalls.py occupancy 60 2.0 > $OUT/r5v_host_stalls_occ_nogclog.txt 2> /dev/null; echo "rc $?"
cat $OUT/r5v_host_stalls_occ.txt | cut -c1-200 | head -120; tail -3 $OUT/r5v_host_stalls_occ.err
