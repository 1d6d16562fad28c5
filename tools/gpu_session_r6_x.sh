#!/bin/bash
# round 6, session x: the decoder layers' assignments in one launch: grounding tests + step A/B
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_grounding.py tests/test_gpu_config4.py tests/test_gpu_optim_table.py tests/test_gpu_insitu.py tests/test_gpu_text_graph.py -m gpu -q -x > $OUT/r6x_tests.txt 2>&1; echo "rc $?"; tail -3 $OUT/r6x_tests.txt
B="python bench.py --no-cpu-baseline --only grounding --steps 10 --warmup 3 --other-steps 10"
for rep in 1 2 3; do
  for v in "ES_MATCH_BATCH=1" "ES_MATCH_BATCH=0"; do
    env $v timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('grounding $v', d['ms_per_step'], d['value'], d['parity']['ok'] if d.get('parity') else None)" | tee -a $OUT/r6x_ab.txt
  done
done
