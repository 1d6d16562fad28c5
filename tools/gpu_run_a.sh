#!/bin/bash
# round-2 GPU session A: full GPU suite, shadow-mode subset, bench (default / resident), rocprof summaries
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -s > gpurun_out/a_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/a_pytest.log
ES_SHADOW=1 timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_config2.py -m gpu -x -q -s > gpurun_out/a_pytest_shadow.log 2>&1; echo "pytest rc=$?" >> gpurun_out/a_pytest_shadow.log
timeout 900 python bench.py > gpurun_out/a_bench.json 2> gpurun_out/a_bench.err
timeout 600 python bench.py --no-cpu-baseline --resident > gpurun_out/a_bench_resident.json 2>> gpurun_out/a_bench.err
ES_SHADOW=1 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/a_bench_shadow.json 2>> gpurun_out/a_bench.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/a_prof -o a -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 8 --warmup 3 > $GRAFT_REPO_ROOT/gpurun_out/a_prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/a_prof.err
cd $GRAFT_REPO_ROOT
ls -la gpurun_out/a_prof | head
DB=$(find gpurun_out/a_prof -name '*.db' | head -1); python tools/rocpd_stats.py $DB gpurun_out/a_kernel_stats.txt > /dev/null 2>&1 || true
find gpurun_out/a_prof -name '*.db' -size +40M -delete
tail -5 gpurun_out/a_pytest.log
cat gpurun_out/a_bench.json | head -c 3000
