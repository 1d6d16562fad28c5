#!/bin/bash
# round 5, session o: where the occupancy neck family's tail goes (every engine launch of a single-stream step) and whether the
# dense weight-gradient kernel's transposed LDS reads conflict
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
db () { find /tmp/prof_$1 -name '*.db' | head -1; }
ES_BENCH_DUMP=$OUT/r5o_occ_launches.jsonl timeout 400 python bench.py --no-cpu-baseline --only occupancy --steps 8 --other-steps 8 --warmup 3 > $OUT/r5o_bench_occ.json 2> $OUT/r5o_bench_occ.err; echo "rc $?"
CMD="python $GRAFT_REPO_ROOT/tools/bench_dconv.py --quick --reps 2"
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d /tmp/prof_lds -o p -- $CMD > /tmp/prof_lds.log 2>&1); echo "rc $?"
python tools/rocpd_pmc.py "$(db lds)" $OUT/r5o_pmc_lds.txt > /dev/null; tail -5 /tmp/prof_lds.log
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY -d /tmp/prof_lds2 -o p -- $CMD > /tmp/prof_lds2.log 2>&1); echo "rc $?"
python tools/rocpd_pmc.py "$(db lds2)" $OUT/r5o_pmc_lds2.txt > /dev/null; tail -5 /tmp/prof_lds2.log
head -30 $OUT/r5o_pmc_lds.txt | cut -c1-200
