#!/bin/bash
# round 5, session d: after the reference-cycle fix -- the three step benches with per-step diagnostics, the from-files leg, the
# main-stream dependent chain of mv-3ddet (kernel trace of the default four-stream schedule + the longest gaps)
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
B="$GRAFT_REPO_ROOT/bench.py"
db () { find /tmp/prof_$1 -name '*.db' | head -1; }
ES_BENCH_DIAG=1 timeout 500 python bench.py --no-cpu-baseline --only grounding --steps 30 --other-steps 30 --warmup 3 > $OUT/r5d_bench_grounding_diag.json 2> $OUT/r5d_bench_grounding_diag.err; echo "rc $?"
timeout 300 python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 > $OUT/r5d_bench_mv3ddet.json 2> $OUT/r5d_bench_mv3ddet.err; echo "rc $?"
timeout 400 python bench.py --no-cpu-baseline --only from_files --steps 20 --other-steps 20 > $OUT/r5d_bench_from_files.json 2> $OUT/r5d_bench_from_files.err; echo "rc $?"
C1="python $B --no-cpu-baseline --no-other-configs --steps 4 --warmup 2"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt_det -o p -- $C1 > /tmp/prof_kt_det.log 2>&1); echo "rc $?"
python tools/rocpd_critical.py "$(db kt_det)" > $OUT/r5d_critical_chain.txt 2>&1
python tools/rocpd_timeline.py "$(db kt_det)" > $OUT/r5d_stream_timeline.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_occ.py tests/test_gpu_ops.py tests/test_gpu_model.py -x -q > $OUT/r5d_tests.txt 2>&1; echo "rc $?"
tail -4 $OUT/r5d_tests.txt
tail -3 $OUT/r5d_bench_from_files.err
