#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
run () { env $1 timeout 300 python bench.py --only $3 --no-cpu-baseline --steps 6 --warmup 2 > $OUT/r4t_$2.json 2> $OUT/r4t_$2.err; echo "== $2 ($1) rc $?"; grep -E "fault|Error|error" $OUT/r4t_$2.err | head -3; python -c "
import json
try:
    d=json.load(open('gpurun_out/r4t_$2.json')); print('   ms', d['ms_per_step'], d['step_ms'], d['engine_all']['launches_per_step'], d['stage_ms'].get('backward (head, 3-D, 2-D)'))
except Exception as e: print('   no json')"; }
run "A=1" g_default grounding
timeout 600 python -m pytest tests/test_gpu_grounding.py tests/test_gpu_insitu.py -q -s -x -p no:cacheprovider -k "not config5_scale and not occupancy" > $OUT/r4t_tests.txt 2>&1
echo "pytest rc $?" >> $OUT/r4t_tests.txt
grep -v Warning $OUT/r4t_tests.txt | grep -E "passed|failed|^E  |FAILED" | head
db () { find /tmp/prof_$1 -name '*.db' | head -1; }
CMD="python $GRAFT_REPO_ROOT/bench.py --only grounding --no-cpu-baseline --steps 3 --warmup 1"
(cd /tmp && ES_TWO_STREAMS=0 ES_WGRAD_ASYNC=0 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_g -o p -- $CMD > /tmp/prof_g.log 2>&1); echo "rc $?"
python tools/rocpd_stats.py "$(db g)" $OUT/r4t_ss_kernel_stats_grounding.txt > /dev/null
grep -v "at::native\|Cijk" $OUT/r4t_ss_kernel_stats_grounding.txt | head -45
