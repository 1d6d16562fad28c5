#!/bin/bash
# round-3 session E: full GPU suite with bf16 activation storage + hand-written radix sort + retuned weight-gradient slices,
# default bench line, A/B of the activation storage
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/r3_e_pytest.txt 2>&1
echo "pytest rc $?" >> gpurun_out/r3_e_pytest.txt
tail -12 gpurun_out/r3_e_pytest.txt
timeout 900 python bench.py > gpurun_out/r3_e_bench.json 2> gpurun_out/r3_e_bench.err
echo "bench rc $?"
tail -c 600 gpurun_out/r3_e_bench.err
head -c 300 gpurun_out/r3_e_bench.json
ES_ACT16=0 timeout 600 python bench.py --no-cpu-baseline --no-other-configs > gpurun_out/r3_e_bench_act16_off.json 2> gpurun_out/r3_e_bench_act16_off.err
echo "bench(act16 off) rc $?"
head -c 300 gpurun_out/r3_e_bench_act16_off.json
