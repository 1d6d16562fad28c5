#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
# parity with the captured image-backbone sequence: run-to-run + config-2 tests replay the graph on their 2nd / 3rd pass
timeout 300 python -m pytest tests/test_gpu_config2.py tests/test_gpu_model.py -m gpu -q -s -x -p no:cacheprovider > gpurun_out/r3t_tests.txt 2>&1
echo "pytest rc $?" >> gpurun_out/r3t_tests.txt
grep -v "^$" gpurun_out/r3t_tests.txt | tail -12
ES_GRAPHS=1 timeout 200 python bench.py --no-cpu-baseline --no-other-configs --steps 8 --warmup 3 > gpurun_out/r3t_bench_graph.json 2> gpurun_out/r3t_bench_graph.err; echo rc $?
python -c "import json;d=json.load(open('gpurun_out/r3t_bench_graph.json'));print('graphs on ', d['value'], d['ms_per_step'], d['step_ms'], d.get('parity',{}).get('ok'))"; tail -3 gpurun_out/r3t_bench_graph.err
ES_GRAPHS=0 timeout 200 python bench.py --no-cpu-baseline --no-other-configs --steps 8 --warmup 3 > gpurun_out/r3t_bench_nograph.json 2> gpurun_out/r3t_bench_nograph.err; echo rc $?
python -c "import json;d=json.load(open('gpurun_out/r3t_bench_nograph.json'));print('graphs off', d['value'], d['ms_per_step'], d['step_ms'], d.get('parity',{}).get('ok'))"
