#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
B="python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 5"
run () { env ES_NEXT_PREFETCH=0 $1 timeout 200 $B > $OUT/r4i_$2.json 2> $OUT/r4i_$2.err; python -c "
import json; d=json.load(open('gpurun_out/r4i_$2.json')); print('$2', '$1', d['ms_per_step'], d['value'])"; }
run "A=1" default
for q in 1 2 3 4; do
  run "GPU_MAX_HW_QUEUES=$q" q${q}
  run "GPU_MAX_HW_QUEUES=$q ES_WGRAD_ASYNC=0" q${q}_nowg
  run "GPU_MAX_HW_QUEUES=$q ES_TWO_STREAMS=0" q${q}_noside
  run "GPU_MAX_HW_QUEUES=$q ES_TWO_STREAMS=0 ES_WGRAD_ASYNC=0" q${q}_single
done
run "A=1" default_again
timeout 300 python -m pytest tests/test_gpu_grounding.py -q -s -p no:cacheprovider -k "train_step" > $OUT/r4i_tests.txt 2>&1
echo "pytest rc $?" >> $OUT/r4i_tests.txt
grep -v Warning $OUT/r4i_tests.txt | grep -E "passed|failed|^E  |FAILED|bf16" | head -40
