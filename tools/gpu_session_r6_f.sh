#!/bin/bash
# round 6 session f: image weight-gradient kernel: A/B tool at the real shapes, default-bench A/B (ES_IMG_WGRAD=1/0), regression tests
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 300 python tools/bench_imgwgrad.py > $OUT/r6f_imgwgrad_ab.txt 2>&1; tail -6 $OUT/r6f_imgwgrad_ab.txt
for h in 1 0; do
  ES_IMG_WGRAD=$h timeout 600 python bench.py --no-other-configs --steps 12 > $OUT/r6f_bench_iw$h.txt 2> $OUT/r6f_bench_iw$h.err; echo "bench rc $?"
  cp bench_detail.json $OUT/r6f_bench_iw${h}_detail.json
  python - <<PY
import json
f=json.load(open('bench_detail.json'))
print('img wgrad $h', f['value'], f['ms_per_step'], f['step_ms'])
print('  ', f['stage_ms'])
print('  ', f['parity']['ok'], f['parity']['rel_err'])
PY
done
timeout 1500 python -m pytest tests/test_gpu_insitu.py tests/test_gpu_resnet2d.py tests/test_gpu_model.py tests/test_gpu_config2.py tests/test_gpu_optim_table.py tests/test_gpu_imgwgrad.py -q -x > $OUT/r6f_tests.txt 2>&1; echo "tests rc $?"
tail -5 $OUT/r6f_tests.txt
for kind in grounding; do
 for h in 1 0; do
  ES_IMG_WGRAD=$h timeout 600 python bench.py --only $kind --no-cpu-baseline --steps 10 > $OUT/r6f_bench_${kind}_iw$h.txt 2>&1
  tail -c 600 $OUT/r6f_bench_${kind}_iw$h.txt
 done
done
