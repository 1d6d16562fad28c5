#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
VIEWS=4 timeout 600 python tools/debug_bf16_grads.py > gpurun_out/i_dbg_default.txt 2>&1
ES_DET_SPLIT=0 VIEWS=4 timeout 600 python tools/debug_bf16_grads.py > gpurun_out/i_dbg_nodet.txt 2>&1
ES_PINGPONG=0 ES_DET_SPLIT=0 VIEWS=4 timeout 600 python tools/debug_bf16_grads.py > gpurun_out/i_dbg_nopp_nodet.txt 2>&1
ES_SHADOW=0 ES_DET_SPLIT=0 VIEWS=4 timeout 600 python tools/debug_bf16_grads.py > gpurun_out/i_dbg_noshadow_nodet.txt 2>&1
for f in gpurun_out/i_dbg_*.txt; do echo == $f; grep -E "^ *[0-9]+ (dY|norm)" $f | awk '{ if ($NF+0 > 0.05) print }' | head -8; tail -3 $f; done
