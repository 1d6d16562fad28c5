#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_draws.py tests/test_gpu_dataset.py -q -s -p no:cacheprovider -k "topk or draws or dataset or device" > $OUT/r4s_tests.txt 2>&1
echo "pytest rc $?" >> $OUT/r4s_tests.txt
grep -v Warning $OUT/r4s_tests.txt | grep -E "passed|failed|^E  |FAILED|device Point|train steps" | head -30
timeout 900 python -X faulthandler bench.py --steps 20 --warmup 5 > $OUT/r4s_bench_default.json 2> $OUT/r4s_bench_default.err; echo "bench rc $?"; tail -5 $OUT/r4s_bench_default.err | cut -c1-300
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4s_bench_default.json'))
print('mv3ddet', d['ms_per_step'], d['value'], 'launches', d['roofline']['launches_per_step'], d['roofline']['frac'], d['roofline'].get('frac_of_binding_roof'))
print('stage', d.get('stage_ms'))
for k,v in d.get('other_configs',{}).items():
    print(k, v.get('ms_per_step'), v.get('value'), v.get('step_ms'), v.get('parity',{}).get('ok'), v.get('error'))
    print('   stage', v.get('stage_ms'))
print('parity', d.get('parity'))
PY
