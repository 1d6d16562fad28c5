#!/bin/bash
# closing session 3 of round 5 (after the FPN's 3x3 convolutions moved to the dense engine: the occupancy path changed, mv-3ddet and
# grounding did not): the default bench line again, and the occupancy kernel trace + PMC passes
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
B="$GRAFT_REPO_ROOT/bench.py"
db () { find /tmp/prof_$1 -name '*.db' | head -1; }
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/r5_bench_default.json 2> $OUT/r5_bench_default.err; echo "bench rc $?"
C2="python $B --no-cpu-baseline --only occupancy --steps 3 --warmup 1 --other-steps 3"
(cd /tmp && ES_TWO_STREAMS=0 ES_WGRAD_ASYNC=0 timeout 250 rocprofv3 --kernel-trace --stats -d /tmp/prof_ks_occupancy -o p -- $C2 > /tmp/prof_ks_occupancy.log 2>&1); echo "rc $?"
python tools/rocpd_stats.py "$(db ks_occupancy)" $OUT/r5_single_stream_kernel_stats_occupancy.txt > /dev/null
(cd /tmp && timeout 250 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY -d /tmp/prof_sq_occupancy -o p -- $C2 > /tmp/prof_sq_occupancy.log 2>&1); echo "rc $?"
python tools/rocpd_pmc.py "$(db sq_occupancy)" $OUT/r5_pmc_sq_occupancy.txt > /dev/null
(cd /tmp && timeout 250 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum -d /tmp/prof_pf_occ -o p -- $C2 > /tmp/prof_pf_occ.log 2>&1); echo "rc $?"
python tools/rocpd_pmc.py "$(db pf_occ)" $OUT/r5_pmc_fetch_occupancy.txt > /dev/null
(cd /tmp && timeout 250 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum -d /tmp/prof_pw_occ -o p -- $C2 > /tmp/prof_pw_occ.log 2>&1); echo "rc $?"
python tools/rocpd_pmc.py "$(db pw_occ)" $OUT/r5_pmc_write_occupancy.txt > /dev/null
ls -la $OUT | grep "r5_" | tail -8
