#!/bin/bash
# round 6 session c: PMC counters of the halo kernel vs the gather kernel on the L0 case (separate passes, kernel trace only)
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
db () { find /tmp/prof_$1 -name '*.db' | head -1; }
CMD="python $GRAFT_REPO_ROOT/tools/bench_halo.py"
(cd /tmp && ONLY_L0=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/prof_h1 -o p -- $CMD > /tmp/prof_h1.log 2>&1); echo "rc $?"
python tools/rocpd_pmc.py "$(db h1)" $OUT/r6c_pmc_halo_sq.txt | grep -i "halo\|spconv_bf16\|kernel |" | cut -c1-400
(cd /tmp && ONLY_L0=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_MISC -d /tmp/prof_h2 -o p -- $CMD > /tmp/prof_h2.log 2>&1); echo "rc $?"
python tools/rocpd_pmc.py "$(db h2)" $OUT/r6c_pmc_halo_sq2.txt | grep -i "halo\|spconv_bf16\|kernel |" | cut -c1-400
tail -3 /tmp/prof_h2.log
