#!/bin/bash
# round 5, session b: kernel traces of the occupancy step (dense engine on) and of mv-3ddet on the current tree, the election stress
# test, the grounding step-time outlier hunt (30 steps with per-step host / allocator diagnostics)
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
B="$GRAFT_REPO_ROOT/bench.py"
db () { find /tmp/prof_$1 -name '*.db' | head -1; }
timeout 600 python -m pytest tests/test_gpu_elect.py -x -q -s > $OUT/r5b_test_elect.txt 2>&1; echo "rc $?"
C2="python $B --no-cpu-baseline --only occupancy --steps 3 --warmup 1 --other-steps 3"
(cd /tmp && ES_TWO_STREAMS=0 ES_WGRAD_ASYNC=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_ks_occ -o p -- $C2 > /tmp/prof_ks_occ.log 2>&1); echo "rc $?"
python tools/rocpd_stats.py "$(db ks_occ)" $OUT/r5b_single_stream_kernel_stats_occupancy.txt > /dev/null
C1="python $B --no-cpu-baseline --no-other-configs --steps 4 --warmup 2"
(cd /tmp && ES_TWO_STREAMS=0 ES_WGRAD_ASYNC=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_ks_det -o p -- $C1 > /tmp/prof_ks_det.log 2>&1); echo "rc $?"
python tools/rocpd_stats.py "$(db ks_det)" $OUT/r5b_single_stream_kernel_stats.txt > /dev/null
ES_BENCH_DIAG=1 timeout 500 python bench.py --no-cpu-baseline --only grounding --steps 30 --other-steps 30 --warmup 3 > $OUT/r5b_bench_grounding_diag.json 2> $OUT/r5b_bench_grounding_diag.err; echo "rc $?"
timeout 300 python bench.py --no-cpu-baseline --only occupancy --steps 8 --other-steps 8 --warmup 3 > $OUT/r5b_bench_occ.json 2> $OUT/r5b_bench_occ.err; echo "rc $?"
tail -3 $OUT/r5b_test_elect.txt
