#!/bin/bash
# round 6 session g: image convolution kernels (forward + gated data gradient): tests, A/B tool, bench A/B (ES_IMG_CONV=1/0), regression
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 400 python -m pytest tests/test_gpu_imgconv.py -q -x -s > $OUT/r6g_test_imgconv.txt 2>&1; echo "imgconv tests rc $?"; tail -14 $OUT/r6g_test_imgconv.txt
timeout 300 python tools/bench_imgconv.py > $OUT/r6g_imgconv_ab.txt 2>&1; tail -7 $OUT/r6g_imgconv_ab.txt
for h in 1 0; do
  ES_IMG_CONV=$h timeout 600 python bench.py --no-other-configs --steps 12 > $OUT/r6g_bench_ic$h.txt 2> $OUT/r6g_bench_ic$h.err; echo "bench rc $?"
  cp bench_detail.json $OUT/r6g_bench_ic${h}_detail.json
  python - <<PY
import json
f=json.load(open('bench_detail.json'))
print('img conv $h', f['value'], f['ms_per_step'], f['step_ms'])
print('  ', {k: v for k, v in f['stage_ms'].items() if k[0] != '_'})
print('  ', f['parity']['ok'], f['parity']['rel_err'])
PY
done
timeout 1500 python -m pytest tests/test_gpu_resnet2d.py tests/test_gpu_insitu.py tests/test_gpu_model.py tests/test_gpu_config2.py tests/test_gpu_optim_table.py tests/test_gpu_prefetch.py -q -x > $OUT/r6g_tests.txt 2>&1; echo "tests rc $?"
tail -5 $OUT/r6g_tests.txt
