#!/bin/bash
# round 5, session m: 128-column tile + sliced parity-class launches (parity, A/B, occupancy step), the default line with ONE copy stream
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_dconv.py -x -q > $OUT/r5m_test_dconv.txt 2>&1; echo "rc $?"
timeout 300 python tools/bench_dconv.py --reps 5 > $OUT/r5m_dconv_ab.txt 2>&1; echo "rc $?"
timeout 300 python bench.py --no-cpu-baseline --only occupancy --steps 10 --other-steps 10 --warmup 3 > $OUT/r5m_bench_occ.json 2> $OUT/r5m_err1.txt; echo "rc $?"
timeout 900 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/r5m_bench_default.json 2> $OUT/r5m_err2.txt; echo "rc $?"
timeout 600 python -m pytest tests/test_gpu_insitu.py tests/test_gpu_occ.py -x -q -k "occ or config5" > $OUT/r5m_tests_occ.txt 2>&1; echo "rc $?"
tail -3 $OUT/r5m_test_dconv.txt $OUT/r5m_tests_occ.txt
grep -E "dgrad" $OUT/r5m_dconv_ab.txt
