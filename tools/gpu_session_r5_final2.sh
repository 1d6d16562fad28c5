#!/bin/bash
# closing session 2 of round 5: the default bench line, kernel traces (four-stream + single-stream) of the three configurations,
# PMC passes (separate passes, kernel trace only) of mv-3ddet and occupancy, the 2-rank gloo plumbing run
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
B="$GRAFT_REPO_ROOT/bench.py"
db () { find /tmp/prof_$1 -name '*.db' | head -1; }
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/r5_bench_default.json 2> $OUT/r5_bench_default.err; echo "bench rc $?"
CMD="python $B --no-cpu-baseline --no-other-configs --steps 4 --warmup 2"
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_ks -o p -- $CMD > /tmp/prof_ks.log 2>&1); echo "rc $?"
python tools/rocpd_stats.py "$(db ks)" $OUT/r5_kernel_stats.txt > /dev/null
python tools/rocpd_critical.py "$(db ks)" > $OUT/r5_critical_chain.txt 2>&1
(cd /tmp && ES_TWO_STREAMS=0 ES_WGRAD_ASYNC=0 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_ss -o p -- $CMD > /tmp/prof_ss.log 2>&1); echo "rc $?"
python tools/rocpd_stats.py "$(db ss)" $OUT/r5_single_stream_kernel_stats.txt > /dev/null
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum -d /tmp/prof_pf -o p -- $CMD > /tmp/prof_pf.log 2>&1); echo "rc $?"
python tools/rocpd_pmc.py "$(db pf)" $OUT/r5_pmc_fetch.txt > /dev/null
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum -d /tmp/prof_pw -o p -- $CMD > /tmp/prof_pw.log 2>&1); echo "rc $?"
python tools/rocpd_pmc.py "$(db pw)" $OUT/r5_pmc_write.txt > /dev/null
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY -d /tmp/prof_sq -o p -- $CMD > /tmp/prof_sq.log 2>&1); echo "rc $?"
python tools/rocpd_pmc.py "$(db sq)" $OUT/r5_pmc_sq.txt > /dev/null
for kind in occupancy grounding; do
  C2="python $B --no-cpu-baseline --only $kind --steps 3 --warmup 1 --other-steps 3"
  (cd /tmp && ES_TWO_STREAMS=0 ES_WGRAD_ASYNC=0 timeout 250 rocprofv3 --kernel-trace --stats -d /tmp/prof_ks_$kind -o p -- $C2 > /tmp/prof_ks_$kind.log 2>&1); echo "rc $?"
  python tools/rocpd_stats.py "$(db ks_$kind)" $OUT/r5_single_stream_kernel_stats_$kind.txt > /dev/null
  (cd /tmp && timeout 250 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY -d /tmp/prof_sq_$kind -o p -- $C2 > /tmp/prof_sq_$kind.log 2>&1); echo "rc $?"
  python tools/rocpd_pmc.py "$(db sq_$kind)" $OUT/r5_pmc_sq_$kind.txt > /dev/null
done
C2="python $B --no-cpu-baseline --only occupancy --steps 3 --warmup 1 --other-steps 3"
(cd /tmp && timeout 250 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum -d /tmp/prof_pf_occ -o p -- $C2 > /tmp/prof_pf_occ.log 2>&1); echo "rc $?"
python tools/rocpd_pmc.py "$(db pf_occ)" $OUT/r5_pmc_fetch_occupancy.txt > /dev/null
(cd /tmp && timeout 250 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum -d /tmp/prof_pw_occ -o p -- $C2 > /tmp/prof_pw_occ.log 2>&1); echo "rc $?"
python tools/rocpd_pmc.py "$(db pw_occ)" $OUT/r5_pmc_write_occupancy.txt > /dev/null
ES_DIST_BACKEND=gloo timeout 400 python bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs > $OUT/r5_bench_2ranks_gloo_one_gpu.json 2> $OUT/r5_2ranks.err; echo "2rank rc $?"
ls -la $OUT | tail -25
