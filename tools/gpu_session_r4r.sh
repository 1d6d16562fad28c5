#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
run () { env $1 timeout 300 python bench.py --only $3 --no-cpu-baseline --steps 4 --warmup 1 > $OUT/r4r_$2.json 2> $OUT/r4r_$2.err; echo "== $2 ($1) rc $?"; grep -E "fault|Error|error" $OUT/r4r_$2.err | head -3; python -c "
import json
try:
    d=json.load(open('gpurun_out/r4r_$2.json')); print('   ms', d['ms_per_step'], d['step_ms'], d['engine_all']['launches_per_step'])
except Exception as e: print('   no json')"; }
run "A=1" g_default grounding
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_grounding.py tests/test_gpu_insitu.py -q -s -x -p no:cacheprovider -k "not config5_scale" > $OUT/r4r_tests.txt 2>&1
echo "pytest rc $?" >> $OUT/r4r_tests.txt
grep -v Warning $OUT/r4r_tests.txt | grep -E "passed|failed|^E  |FAILED" | head -30
timeout 900 python -X faulthandler bench.py --steps 20 --warmup 5 > $OUT/r4r_bench_default.json 2> $OUT/r4r_bench_default.err; echo "bench rc $?"; tail -25 $OUT/r4r_bench_default.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4r_bench_default.json'))
print('mv3ddet', d['ms_per_step'], d['value'], 'launches', d['roofline']['launches_per_step'], d['roofline']['frac'], d['roofline'].get('frac_of_binding_roof'))
print('stage', d.get('stage_ms'))
for k,v in d.get('other_configs',{}).items():
    print(k, v.get('ms_per_step'), v.get('value'), v.get('step_ms'), v.get('parity',{}).get('ok'), v.get('error'))
    print('   stage', v.get('stage_ms'))
print('parity', d.get('parity'))
PY
