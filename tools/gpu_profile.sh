#!/bin/bash
# Regenerates the round-3 files under profiles/ (kernel traces, PMC passes, bench lines).  Run through gpurun:
#   gpurun --timeout 3000 -- 'bash tools/gpu_profile.sh'      (outputs land in gpurun_out/, copy the summaries to profiles/)
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
B="$GRAFT_REPO_ROOT/bench.py"
run_prof () {   # name, extra rocprof args..., then -- command
  local name=$1; shift
  rm -rf /tmp/prof_$name
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace "$@" > /tmp/prof_$name.log 2>&1)
  echo "rocprof $name rc $?"
}
db () { find /tmp/prof_$1 -name '*.db' | head -1; }
CMD="python $B --no-cpu-baseline --no-other-configs --steps 4 --warmup 2"
# 1. kernel trace, default four-stream schedule and single-stream schedule
run_prof ks --stats -d /tmp/prof_ks -o p -- $CMD
python tools/rocpd_stats.py "$(db ks)" $OUT/r3_kernel_stats.txt > /dev/null
ES_TWO_STREAMS=0 ES_WGRAD_ASYNC=0 run_prof ss --stats -d /tmp/prof_ss -o p -- $CMD
python tools/rocpd_stats.py "$(db ss)" $OUT/r3_single_stream_kernel_stats.txt > /dev/null
# 2. PMC passes of the same command (separate passes, no other trace domains)
run_prof pf --pmc FETCH_SIZE TCC_HIT_sum -d /tmp/prof_pf -o p -- $CMD
python tools/rocpd_pmc.py "$(db pf)" $OUT/r3_pmc_fetch.txt > /dev/null
run_prof pw --pmc WRITE_SIZE TCC_MISS_sum -d /tmp/prof_pw -o p -- $CMD
python tools/rocpd_pmc.py "$(db pw)" $OUT/r3_pmc_write.txt > /dev/null
run_prof sq --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY -d /tmp/prof_sq -o p -- $CMD
python tools/rocpd_pmc.py "$(db sq)" $OUT/r3_pmc_sq.txt > /dev/null
# 3. configs 4 and 5: traffic of their named families (attention / conv engine)
for kind in grounding occupancy; do
  C2="python $B --no-cpu-baseline --only $kind --steps 3 --warmup 2 --other-steps 3"
  run_prof f_$kind --pmc FETCH_SIZE TCC_HIT_sum -d /tmp/prof_f_$kind -o p -- $C2
  python tools/rocpd_pmc.py "$(db f_$kind)" $OUT/r3_pmc_fetch_$kind.txt > /dev/null
  run_prof w_$kind --pmc WRITE_SIZE TCC_MISS_sum -d /tmp/prof_w_$kind -o p -- $C2
  python tools/rocpd_pmc.py "$(db w_$kind)" $OUT/r3_pmc_write_$kind.txt > /dev/null
done
run_prof sq_grounding --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY -d /tmp/prof_sq_grounding -o p -- python $B --no-cpu-baseline --only grounding --steps 3 --warmup 2 --other-steps 3
python tools/rocpd_pmc.py "$(db sq_grounding)" $OUT/r3_pmc_sq_grounding.txt > /dev/null
ls -la $OUT | tail -20
