#!/bin/bash
# round-3 session A: full GPU suite (incl. the config-4 / config-5 scale tests) + the default bench line
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/r3_a_pytest.txt 2>&1
echo "pytest rc $?" >> gpurun_out/r3_a_pytest.txt
tail -5 gpurun_out/r3_a_pytest.txt
timeout 900 python bench.py > gpurun_out/r3_a_bench.json 2> gpurun_out/r3_a_bench.err
echo "bench rc $?"
tail -c 1500 gpurun_out/r3_a_bench.err
head -c 600 gpurun_out/r3_a_bench.json
