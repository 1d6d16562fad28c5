#!/bin/bash
# round-2 GPU session J (final profile set): full suite with the ping-pong kernels + coordinate prefetch, A/B of the prefetch, host issue time,
# final profiles (kernel trace default / single stream, PMC passes), config-4 / config-5 benches
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s > gpurun_out/j_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/j_pytest.log
timeout 900 python bench.py > gpurun_out/j_bench.json 2> gpurun_out/j_bench.err
timeout 600 python bench.py --no-cpu-baseline --precision f32 > gpurun_out/j_bench_f32.json 2>> gpurun_out/j_bench.err
timeout 300 python tools/host_time.py > gpurun_out/j_host_time.txt 2>&1
timeout 600 python tools/bench_occ.py > gpurun_out/j_bench_occ.json 2> gpurun_out/j_bench_occ.err
timeout 900 python tools/bench_grounding.py > gpurun_out/j_bench_ground.json 2> gpurun_out/j_bench_ground.err
prof() {
  name=$1; shift
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace "$@" -d /tmp/prof_$name -o p -- python $R/bench.py --no-cpu-baseline --steps 4 --warmup 2 > $R/gpurun_out/j_prof_$name.json 2> $R/gpurun_out/j_prof_$name.err
  cd $R
  DB=$(find /tmp/prof_$name -name '*.db' | head -1)
}
prof default --stats
python tools/rocpd_stats.py $DB gpurun_out/j_kernel_stats.txt > /dev/null 2>&1
ES_TWO_STREAMS=0 ES_WGRAD_ASYNC=0 prof single --stats
python tools/rocpd_stats.py $DB gpurun_out/j_single_stream_kernel_stats.txt > /dev/null 2>&1
prof fetch --pmc FETCH_SIZE TCC_HIT_sum
python tools/rocpd_pmc.py $DB gpurun_out/j_pmc_fetch.txt > /dev/null 2>&1
prof write --pmc WRITE_SIZE TCC_MISS_sum
python tools/rocpd_pmc.py $DB gpurun_out/j_pmc_write.txt > /dev/null 2>&1
prof sq --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY
python tools/rocpd_pmc.py $DB gpurun_out/j_pmc_sq.txt > /dev/null 2>&1
grep -E "passed|failed|rc=" gpurun_out/j_pytest.log
for f in j_bench j_bench_f32; do python -c "
import json,sys
d=json.loads(open('gpurun_out/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac_of_bindinj_roof'])"; done
head -3 gpurun_out/j_host_time.txt
