#!/bin/bash
# round-3 closing session: the default bench line with the final bench.py, and the FETCH / WRITE passes again with the full
# per-kernel table (the first tables were cut at 60 kernels, which dropped the small scatter-path kernels)
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p $OUT
timeout 900 python bench.py > $OUT/r3_bench_default.json 2> $OUT/r3_bench_default.err
echo "bench rc $?"; head -c 300 $OUT/r3_bench_default.json
B="$GRAFT_REPO_ROOT/bench.py"
CMD="python $B --no-cpu-baseline --no-other-configs --steps 4 --warmup 2"
db () { find /tmp/prof_$1 -name '*.db' | head -1; }
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum -d /tmp/prof_pf -o p -- $CMD > /tmp/prof_pf.log 2>&1); echo "pf rc $?"
python tools/rocpd_pmc.py "$(db pf)" $OUT/r3_pmc_fetch.txt > /dev/null
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum -d /tmp/prof_pw -o p -- $CMD > /tmp/prof_pw.log 2>&1); echo "pw rc $?"
python tools/rocpd_pmc.py "$(db pw)" $OUT/r3_pmc_write.txt > /dev/null
wc -l $OUT/r3_pmc_fetch.txt $OUT/r3_pmc_write.txt
