#!/bin/bash
# closing session 3 of round 4: the remaining GPU tests on the final tree (part 1 of the suite), then, budget permitting, the PMC
# traffic passes of the other two configurations
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_dma.py tests/test_gpu_experimental.py tests/test_gpu_model.py tests/test_gpu_grounding.py tests/test_gpu_insitu.py tests/test_gpu_draws.py -q -s -p no:cacheprovider > $OUT/r4_gputest_part1.txt 2>&1
echo "pytest rc $?" >> $OUT/r4_gputest_part1.txt
grep -v Warning $OUT/r4_gputest_part1.txt | grep -E "passed|failed|^E  |FAILED" | head
B="$GRAFT_REPO_ROOT/bench.py"
db () { find /tmp/prof_$1 -name '*.db' | head -1; }
for kind in occupancy grounding; do
  C2="python $B --no-cpu-baseline --only $kind --steps 3 --warmup 1 --other-steps 3"
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum -d /tmp/prof_pf_$kind -o p -- $C2 > /tmp/prof_pf_$kind.log 2>&1); echo "rc $?"
  python tools/rocpd_pmc.py "$(db pf_$kind)" $OUT/r4_pmc_fetch_$kind.txt > /dev/null
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum -d /tmp/prof_pw_$kind -o p -- $C2 > /tmp/prof_pw_$kind.log 2>&1); echo "rc $?"
  python tools/rocpd_pmc.py "$(db pw_$kind)" $OUT/r4_pmc_write_$kind.txt > /dev/null
done
