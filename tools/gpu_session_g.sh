#!/bin/bash
# round-3 final session: full GPU suite + smoke, the default bench line, then every profile under profiles/ (tools/gpu_profile.sh)
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/r3_gputest_full.txt 2>&1
echo "pytest rc $?" >> gpurun_out/r3_gputest_full.txt
tail -6 gpurun_out/r3_gputest_full.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3_smoke.txt 2>&1
echo "smoke rc $?"; tail -4 gpurun_out/r3_smoke.txt
timeout 900 python bench.py > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err
echo "bench rc $?"
tail -c 400 gpurun_out/r3_bench_default.err
head -c 300 gpurun_out/r3_bench_default.json
bash tools/gpu_profile.sh > gpurun_out/r3_profile.log 2>&1
echo "profile rc $?"
tail -25 gpurun_out/r3_profile.log
