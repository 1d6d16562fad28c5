#!/bin/bash
# round-2 GPU session E: full GPU suite, the three benches, kernel-trace summaries (default + single-stream), PMC passes
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
python -m pytest tests -m gpu -q -s > gpurun_out/e_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/e_pytest.log
timeout 900 python bench.py > gpurun_out/e_bench.json 2> gpurun_out/e_bench.err
timeout 600 python tools/bench_occ.py > gpurun_out/e_bench_occ.json 2> gpurun_out/e_bench_occ.err
timeout 900 python tools/bench_grounding.py > gpurun_out/e_bench_ground.json 2> gpurun_out/e_bench_ground.err
prof() {   # name, env, extra rocprof args...
  name=$1; shift
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace "$@" -d /tmp/prof_$name -o p -- python $R/bench.py --no-cpu-baseline --steps 4 --warmup 2 > $R/gpurun_out/e_prof_$name.json 2> $R/gpurun_out/e_prof_$name.err
  cd $R
  DB=$(find /tmp/prof_$name -name '*.db' | head -1)
}
prof default --stats
python tools/rocpd_stats.py $DB gpurun_out/e_kernel_stats.txt > /dev/null 2>&1
ES_TWO_STREAMS=0 ES_WGRAD_ASYNC=0 prof single --stats
python tools/rocpd_stats.py $DB gpurun_out/e_single_stream_kernel_stats.txt > /dev/null 2>&1
prof fetch --pmc FETCH_SIZE TCC_HIT_sum
python tools/rocpd_pmc.py $DB gpurun_out/e_pmc_fetch.txt > /dev/null 2>&1
prof write --pmc WRITE_SIZE TCC_MISS_sum
python tools/rocpd_pmc.py $DB gpurun_out/e_pmc_write.txt > /dev/null 2>&1
prof sq --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY
python tools/rocpd_pmc.py $DB gpurun_out/e_pmc_sq.txt > /dev/null 2>&1
grep -E "passed|failed|rc=" gpurun_out/e_pytest.log
head -c 600 gpurun_out/e_bench.json; echo; head -c 900 gpurun_out/e_bench_occ.json; echo; head -c 1500 gpurun_out/e_bench_ground.json
