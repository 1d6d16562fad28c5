#!/bin/bash
# round 6, session l: stem + pool tile variants; graph replay of the frozen text encoder (probe, then grounding A/B)
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 300 python -m pytest tests/test_gpu_imgconv.py -m gpu -q -s -x -k 'stem' > $OUT/r6l_stem_tests.txt 2>&1; echo "rc $?"; grep 'stem + pool\|passed\|failed' $OUT/r6l_stem_tests.txt
timeout 200 python tools/probe_text_graph.py > $OUT/r6l_text_graph_probe.txt 2>&1; echo "rc $?"; tail -12 $OUT/r6l_text_graph_probe.txt
timeout 900 python -m pytest tests/test_gpu_config4.py tests/test_gpu_grounding.py tests/test_gpu_optim_table.py -m gpu -q -x > $OUT/r6l_tests.txt 2>&1; echo "rc $?"; tail -3 $OUT/r6l_tests.txt
B="python bench.py --no-cpu-baseline --only grounding --steps 10 --warmup 3 --other-steps 10"
for rep in 1 2 3; do
  for v in "ES_TEXT_GRAPH=1" "ES_TEXT_GRAPH=0"; do
    env $v timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['value'])" | tee -a $OUT/r6l_ab.txt
  done
done
