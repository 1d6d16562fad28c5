#!/bin/bash
# round 6, session o: few-row linear kernels + expansion stream kernel: tests, then grounding / mv-3ddet step A/B
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_linear.py -m gpu -q -s > $OUT/r6o_linear_tests.txt 2>&1; echo "rc $?"; grep 'kernel\|passed\|failed' $OUT/r6o_linear_tests.txt | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_config4.py tests/test_gpu_grounding.py tests/test_gpu_optim_table.py tests/test_gpu_config2.py tests/test_gpu_resnet2d.py tests/test_gpu_insitu.py -m gpu -q -x > $OUT/r6o_tests.txt 2>&1; echo "rc $?"; tail -3 $OUT/r6o_tests.txt
B="python bench.py --no-cpu-baseline --only grounding --steps 10 --warmup 3 --other-steps 10"
for rep in 1 2 3; do
  for v in "ES_EXPAND=65536" "ES_EXPAND=0"; do
    env $v timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('grounding $v', d['ms_per_step'], d['value']); import os; dd=json.load(open('bench_detail_grounding.json')) if os.path.exists('bench_detail_grounding.json') else {}; print('   step_ms', dd.get('step_ms'))" | tee -a $OUT/r6o_ab.txt
  done
done
B="python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 5"
for rep in 1 2; do
  for v in "ES_EXPAND=65536" "ES_EXPAND=0"; do
    env $v timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mv3ddet $v', d['ms_per_step'], d['value'])" | tee -a $OUT/r6o_ab.txt
  done
done
