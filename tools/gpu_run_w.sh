#!/bin/bash
# round-2 final GPU session: PMC passes -> traffic json, full suite, default / f32 / config-4 / config-5 benches, kernel traces
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
prof() {
  name=$1; shift
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace "$@" -d /tmp/prof_$name -o p -- python $R/bench.py --no-cpu-baseline --steps 4 --warmup 2 > $R/gpurun_out/w_prof_$name.json 2> $R/gpurun_out/w_prof_$name.err
  cd $R
  DB=$(find /tmp/prof_$name -name '*.db' | head -1)
}
prof fetch --pmc FETCH_SIZE TCC_HIT_sum
python tools/rocpd_pmc.py $DB gpurun_out/w_pmc_fetch.txt > /dev/null 2>&1
prof write --pmc WRITE_SIZE TCC_MISS_sum
python tools/rocpd_pmc.py $DB gpurun_out/w_pmc_write.txt > /dev/null 2>&1
cp gpurun_out/w_pmc_fetch.txt profiles/r2_pmc_fetch.txt; cp gpurun_out/w_pmc_write.txt profiles/r2_pmc_write.txt
python tools/pmc_traffic.py > gpurun_out/w_pmc_traffic.log 2>&1; cp profiles/r2_pmc_traffic.json gpurun_out/w_pmc_traffic.json
prof default --stats
python tools/rocpd_stats.py $DB gpurun_out/w_kernel_stats.txt > /dev/null 2>&1
ES_TWO_STREAMS=0 ES_WGRAD_ASYNC=0 prof single --stats
python tools/rocpd_stats.py $DB gpurun_out/w_single_stream_kernel_stats.txt > /dev/null 2>&1
timeout 900 python bench.py > gpurun_out/w_bench.json 2> gpurun_out/w_bench.err
timeout 600 python bench.py --no-cpu-baseline --precision f32 > gpurun_out/w_bench_f32.json 2>> gpurun_out/w_bench.err
timeout 600 python tools/bench_occ.py > gpurun_out/w_bench_occ.json 2> gpurun_out/w_bench_occ.err
timeout 900 python tools/bench_grounding.py > gpurun_out/w_bench_ground.json 2> gpurun_out/w_bench_ground.err
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/w_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/w_smoke.log
python -m pytest tests -m gpu -q -s > gpurun_out/w_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/w_pytest.log
set +x
grep -E "passed|failed|rc=" gpurun_out/w_pytest.log; tail -2 gpurun_out/w_smoke.log
cat gpurun_out/w_pmc_traffic.log
for f in w_bench w_bench_f32 w_bench_occ w_bench_ground; do python -c "
import json
d=json.loads(open('gpurun_out/$f.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$f', d['value'], d['ms_per_step'], r.get('frac'), r.get('frac_of_binding_roof'), r.get('traffic'))"; done
