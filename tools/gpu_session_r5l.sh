#!/bin/bash
# round 5, session l: what the head forward / losses stage and the backward of an mv-3ddet step are made of (single-stream kernel trace, windows)
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
B="$GRAFT_REPO_ROOT/bench.py"
db () { find /tmp/prof_$1 -name '*.db' | head -1; }
C1="python $B --no-cpu-baseline --no-other-configs --steps 4 --warmup 2"
(cd /tmp && ES_TWO_STREAMS=0 ES_WGRAD_ASYNC=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_ss -o p -- $C1 > /tmp/prof_ss.log 2>&1); echo "rc $?"
python tools/rocpd_window.py "$(db ss)" k_point_sample_fwd 'k_pos_losses$' > $OUT/r5l_window_head.txt 2>&1
python tools/rocpd_window.py "$(db ss)" k_pos_losses k_sumsq > $OUT/r5l_window_backward.txt 2>&1
python tools/rocpd_window.py "$(db ss)" k_preprocess_img k_point_sample_fwd > $OUT/r5l_window_front.txt 2>&1
cat $OUT/r5l_window_head.txt
