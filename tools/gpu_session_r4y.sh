#!/bin/bash
# kernel traces + SQ counters of the other two configurations (final code)
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
B="$GRAFT_REPO_ROOT/bench.py"
db () { find /tmp/prof_$1 -name '*.db' | head -1; }
for kind in occupancy grounding; do
  C2="python $B --no-cpu-baseline --only $kind --steps 3 --warmup 1 --other-steps 3"
  (cd /tmp && ES_TWO_STREAMS=0 ES_WGRAD_ASYNC=0 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_ks_$kind -o p -- $C2 > /tmp/prof_ks_$kind.log 2>&1); echo "rc $?"
  python tools/rocpd_stats.py "$(db ks_$kind)" $OUT/r4_single_stream_kernel_stats_$kind.txt > /dev/null
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY -d /tmp/prof_sq_$kind -o p -- $C2 > /tmp/prof_sq_$kind.log 2>&1); echo "rc $?"
  python tools/rocpd_pmc.py "$(db sq_$kind)" $OUT/r4_pmc_sq_$kind.txt > /dev/null
done
