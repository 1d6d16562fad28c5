#!/bin/bash
# kernel-level A/B of the late round-3 changes: stand-alone (single-stream) kernel traces with the LDS-DMA kernel off / 32-channel /
# 64-channel chunks, for the mv-3ddet step and the occupancy step, plus one SQ/LDS counter pass each for off and on
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
B="$GRAFT_REPO_ROOT/bench.py"
db () { find /tmp/prof_$1 -name '*.db' | head -1; }
export ES_TWO_STREAMS=0 ES_WGRAD_ASYNC=0
CMD="python $B --no-cpu-baseline --no-other-configs --steps 4 --warmup 2"
OCC="python $B --no-cpu-baseline --only occupancy --steps 3 --warmup 2 --other-steps 3"
for d in 0 1 2; do
  rm -rf /tmp/prof_m$d
  (cd /tmp && ES_DMA=$d timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_m$d -o p -- $CMD > /tmp/prof_m$d.log 2>&1)
  echo "rc $?"
  python tools/rocpd_stats.py "$(db m$d)" $OUT/r3k_ss_kernel_stats_dma$d.txt > /dev/null
done
for d in 0 2; do
  rm -rf /tmp/prof_o$d
  (cd /tmp && ES_DMA=$d timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_o$d -o p -- $OCC > /tmp/prof_o$d.log 2>&1)
  echo "rc $?"
  python tools/rocpd_stats.py "$(db o$d)" $OUT/r3k_ss_kernel_stats_occ_dma$d.txt > /dev/null
done
for d in 0 1; do
  rm -rf /tmp/prof_q$d
  (cd /tmp && ES_DMA=$d timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES -d /tmp/prof_q$d -o p -- $CMD > /tmp/prof_q$d.log 2>&1)
  echo "rc $?"; tail -2 /tmp/prof_q$d.log
  python tools/rocpd_pmc.py "$(db q$d)" $OUT/r3k_pmc_lds_dma$d.txt > /dev/null
done
ls -la $OUT | tail -12
