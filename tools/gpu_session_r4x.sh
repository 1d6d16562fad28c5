#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 400 python -m pytest tests/test_gpu_insitu.py tests/test_gpu_grounding.py -q -s -x -p no:cacheprovider -k "specification or train_step" > $OUT/r4x_tests.txt 2>&1
echo "pytest rc $?" >> $OUT/r4x_tests.txt
grep -v Warning $OUT/r4x_tests.txt | grep -E "passed|failed|^E  |FAILED" | head
timeout 200 python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 > $OUT/r4x_bench.json 2> $OUT/r4x_bench.err; echo "rc $?"
python -c "
import json; d=json.load(open('gpurun_out/r4x_bench.json')); print('bench', d['ms_per_step'], d['value'], d['roofline']['traffic_source'], d['roofline']['traffic'], d['roofline']['traffic_launches_per_step'])"
timeout 200 python bench.py --only grounding --no-cpu-baseline --steps 6 --warmup 2 > $OUT/r4x_g.json 2> $OUT/r4x_g.err; python -c "
import json; d=json.load(open('gpurun_out/r4x_g.json')); print('grounding', d['ms_per_step'], d['step_ms'])"
