#!/bin/bash
# round 6 session h: map stream (kernel / inverse maps off the dependent chain): bench A/B (ES_MAP_ASYNC=1/0), regression, A/B tools for the record
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
for h in 1 0 1 0; do
  ES_MAP_ASYNC=$h timeout 600 python bench.py --no-other-configs --no-cpu-baseline --steps 16 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('map async $h', d['value'], d['ms_per_step'])"
done
for h in 1 0; do
  ES_MAP_ASYNC=$h timeout 600 python bench.py --only grounding --no-cpu-baseline --steps 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('grounding map async $h', d['value'], d['ms_per_step'])"
done
timeout 1500 python -m pytest tests/test_gpu_config2.py tests/test_gpu_model.py tests/test_gpu_insitu.py tests/test_gpu_ops.py tests/test_gpu_config4.py tests/test_gpu_grounding.py tests/test_gpu_optim_table.py tests/test_gpu_prefetch.py tests/test_gpu_predict.py -q -x > $OUT/r6h_tests.txt 2>&1; echo "tests rc $?"
tail -4 $OUT/r6h_tests.txt
timeout 300 python tools/bench_imgwgrad.py > $OUT/r6h_imgwgrad_ab.txt 2>&1
