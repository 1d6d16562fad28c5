#!/bin/bash
# round 6 session d: halo kernel v2 (half-step pipeline, persistent workgroups, coalesced plan, mirrored data gradient) + children-first
# union order: halo tests, in-situ halo statistics, A/B tool, default-bench A/B (ES_HALO=1/0), config-2 / in-situ / model tests
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_halo.py -q -s > $OUT/r6d_test_halo.txt 2>&1; echo "halo tests rc $?"
tail -4 $OUT/r6d_test_halo.txt
timeout 300 python tools/halo_stats.py > $OUT/r6d_halo_stats.txt 2>&1; tail -8 $OUT/r6d_halo_stats.txt
timeout 600 python tools/bench_halo.py > $OUT/r6d_halo_ab.txt 2>&1; tail -8 $OUT/r6d_halo_ab.txt
for h in 1 0; do
  ES_HALO=$h timeout 600 python bench.py --no-other-configs --steps 12 > $OUT/r6d_bench_halo$h.txt 2> $OUT/r6d_bench_halo$h.err; echo "bench rc $?"
  cp bench_detail.json $OUT/r6d_bench_halo${h}_detail.json
  python - <<PY
import json
f=json.load(open('bench_detail.json'))
print('halo $h', f['value'], f['ms_per_step'], f['step_ms'])
for c in f['roofline']['classes'][:4]: print('  ', c['cls'], c['launches'], c['ms'], c['tflops'])
print('  ', f['parity']['ok'], f['parity']['rel_err'])
PY
done
timeout 1500 python -m pytest tests/test_gpu_config2.py tests/test_gpu_insitu.py tests/test_gpu_model.py tests/test_gpu_ops.py tests/test_gpu_grounding.py tests/test_gpu_config4.py tests/test_gpu_predict.py -q -x > $OUT/r6d_tests.txt 2>&1; echo "tests rc $?"
tail -5 $OUT/r6d_tests.txt
