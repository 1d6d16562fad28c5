#!/bin/bash
# closing session 1 of round 4: the GPU tests not re-run since the last kernel changes, smoke, the default bench line, kernel traces
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_config2.py tests/test_gpu_config4.py tests/test_gpu_config5.py tests/test_gpu_occ.py tests/test_gpu_predict.py tests/test_gpu_resnet2d.py tests/test_gpu_fusion_losses.py tests/test_gpu_zz_rccl.py tests/test_gpu_prefetch.py tests/test_gpu_configs.py tests/test_gpu_dataset.py -q -s -p no:cacheprovider > $OUT/r4_gputest_part2.txt 2>&1
echo "pytest rc $?" >> $OUT/r4_gputest_part2.txt
grep -v Warning $OUT/r4_gputest_part2.txt | grep -E "passed|failed|^E  |FAILED" | head
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/r4_smoke.txt 2>&1
echo "smoke rc $?"; tail -2 $OUT/r4_smoke.txt
timeout 500 python bench.py > $OUT/r4_bench_default.json 2> $OUT/r4_bench_default.err
echo "bench rc $?"
tail -c 300 $OUT/r4_bench_default.err
head -c 300 $OUT/r4_bench_default.json
B="$GRAFT_REPO_ROOT/bench.py"
db () { find /tmp/prof_$1 -name '*.db' | head -1; }
CMD="python $B --no-cpu-baseline --no-other-configs --steps 4 --warmup 2"
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/prof_ks -o p -- $CMD > /tmp/prof_ks.log 2>&1); echo "rc $?"
python tools/rocpd_stats.py "$(db ks)" $OUT/r4_kernel_stats.txt > /dev/null
python tools/rocpd_timeline.py "$(db ks)" 8 > $OUT/r4_stream_timeline.txt 2>&1
python tools/rocpd_critical.py "$(db ks)" 8 > $OUT/r4_critical_chain.txt 2>&1
