#!/bin/bash
# round 4, session a: the three never-run experimental paths (VERDICT r3 "weak" 3) + the transposed-read probe + A/B on the step
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/probes/tr_read.hip -o /tmp/tr_read && timeout 60 /tmp/tr_read > $OUT/r4a_tr_read.txt 2>&1
echo "probe rc $?"
ES_TEST_EXPERIMENTAL=1 timeout 300 python -m pytest tests/test_gpu_experimental.py -q -s -p no:cacheprovider > $OUT/r4a_experimental.txt 2>&1
echo "pytest rc $?" >> $OUT/r4a_experimental.txt
tail -30 $OUT/r4a_experimental.txt
timeout 300 python tools/sweep_options.py --steps 10 --warmup 3 --variants "GEN_FUSED=1;14=1;10=3,11=0" > $OUT/r4a_sweep.txt 2> $OUT/r4a_sweep.err
echo "sweep rc $?"; cat $OUT/r4a_sweep.txt; tail -5 $OUT/r4a_sweep.err
