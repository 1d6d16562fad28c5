#!/bin/bash
# round 5, session w: what the host threads do during the slow occupancy steps (watchdog), and two A/B runs: one intra-op thread,
# no per-step host->device copy
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
HS_GC_LOG=0 HS_WATCH=1 timeout 300 python tools/host_stalls.py occupancy 60 2.0 > $OUT/r5w_watch.txt 2> $OUT/r5w_watch.err; echo "rc $?"
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
HS_GC_LOG=0 HS_THREADS1=1 timeout 300 python tools/host_stalls.py occupancy 60 8.0 > $OUT/r5w_threads1.txt 2> /dev/null; echo "rc $?"
HS_GC_LOG=0 HS_RESIDENT=1 timeout 300 python tools/host_stalls.py occupancy 60 8.0 > $OUT/r5w_resident.txt 2> /dev/null; echo "rc $?"
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
grep -h "steps, median\|threads" $OUT/r5w_watch.txt $OUT/r5w_threads1.txt $OUT/r5w_resident.txt
grep -c "slow" $OUT/r5w_watch.txt $OUT/r5w_threads1.txt $OUT/r5w_resident.txt
sed -n '/^watchdog/,$p' $OUT/r5w_watch.txt | cut -c1-900 | head -70
