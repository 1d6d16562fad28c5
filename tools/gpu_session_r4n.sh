#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_insitu.py tests/test_gpu_config2.py tests/test_gpu_predict.py -q -s -x -p no:cacheprovider -k "not config4_scale and not config5_scale" > $OUT/r4n_tests.txt 2>&1
echo "pytest rc $?" >> $OUT/r4n_tests.txt
grep -v Warning $OUT/r4n_tests.txt | grep -E "passed|failed|^E  |FAILED|^mv-|^occ" | head -30
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/r4n_bench_default.json 2> $OUT/r4n_bench_default.err; echo "bench rc $?"; tail -3 $OUT/r4n_bench_default.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4n_bench_default.json'))
print('mv3ddet', d['ms_per_step'], d['value'], 'launches', d['roofline']['launches_per_step'], d['roofline']['frac'], d['roofline'].get('frac_of_binding_roof'))
print('stage', d.get('stage_ms'))
for k,v in d.get('other_configs',{}).items():
    print(k, v['ms_per_step'], v['value'], v['step_ms'], v.get('parity',{}).get('ok'), v['engine_all']['launches_per_step'])
    print('   stage', v['stage_ms'])
print('parity', d.get('parity'))
PY
