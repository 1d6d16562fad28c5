#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_dma.py tests/test_gpu_experimental.py tests/test_gpu_model.py tests/test_gpu_config2.py tests/test_gpu_fusion_losses.py -q -s -x -p no:cacheprovider > $OUT/r4k_tests.txt 2>&1
echo "pytest rc $?" >> $OUT/r4k_tests.txt
grep -v Warning $OUT/r4k_tests.txt | grep -E "passed|failed|^E  |FAILED" | head -30
B="python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 5"
timeout 200 $B > $OUT/r4k_bench.json 2> $OUT/r4k_bench.err; echo "rc $?"; tail -3 $OUT/r4k_bench.err
python -c "
import json; d=json.load(open('gpurun_out/r4k_bench.json')); print('bench', d['ms_per_step'], d['value'], d['roofline']['launches_per_step'], d.get('parity'))"
db () { find /tmp/prof_$1 -name '*.db' | head -1; }
CMD="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-other-configs --steps 4 --warmup 3"
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/prof_ks -o p -- $CMD > /tmp/prof_ks.log 2>&1); echo "rc $?"
python tools/rocpd_stats.py "$(db ks)" $OUT/r4k_kernel_stats.txt > /dev/null
python tools/rocpd_critical.py "$(db ks)" 3 > $OUT/r4k_critical.txt 2>&1
head -70 $OUT/r4k_critical.txt
