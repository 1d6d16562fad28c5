#!/bin/bash
# PMC passes of the final round-3 code (separate passes, kernel trace only): HBM fetch / write bytes and the SQ counters
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
B="$GRAFT_REPO_ROOT/bench.py"
db () { find /tmp/prof_$1 -name '*.db' | head -1; }
CMD="python $B --no-cpu-baseline --no-other-configs --steps 4 --warmup 2"
(cd /tmp && timeout 100 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum -d /tmp/prof_pf -o p -- $CMD > /tmp/prof_pf.log 2>&1); echo "rc $?"
python tools/rocpd_pmc.py "$(db pf)" $OUT/r3_pmc_fetch.txt > /dev/null
(cd /tmp && timeout 100 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum -d /tmp/prof_pw -o p -- $CMD > /tmp/prof_pw.log 2>&1); echo "rc $?"
python tools/rocpd_pmc.py "$(db pw)" $OUT/r3_pmc_write.txt > /dev/null
(cd /tmp && timeout 100 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY -d /tmp/prof_sq -o p -- $CMD > /tmp/prof_sq.log 2>&1); echo "rc $?"
python tools/rocpd_pmc.py "$(db sq)" $OUT/r3_pmc_sq.txt > /dev/null
ls -la $OUT | tail -5
