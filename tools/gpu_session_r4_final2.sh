#!/bin/bash
# closing session 2 of round 4: PMC passes of the final code (separate passes, kernel trace only), single-stream trace, 2-rank gloo bench
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
B="$GRAFT_REPO_ROOT/bench.py"
db () { find /tmp/prof_$1 -name '*.db' | head -1; }
CMD="python $B --no-cpu-baseline --no-other-configs --steps 4 --warmup 2"
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum -d /tmp/prof_pf -o p -- $CMD > /tmp/prof_pf.log 2>&1); echo "rc $?"
python tools/rocpd_pmc.py "$(db pf)" $OUT/r4_pmc_fetch.txt > /dev/null
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum -d /tmp/prof_pw -o p -- $CMD > /tmp/prof_pw.log 2>&1); echo "rc $?"
python tools/rocpd_pmc.py "$(db pw)" $OUT/r4_pmc_write.txt > /dev/null
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY -d /tmp/prof_sq -o p -- $CMD > /tmp/prof_sq.log 2>&1); echo "rc $?"
python tools/rocpd_pmc.py "$(db sq)" $OUT/r4_pmc_sq.txt > /dev/null
(cd /tmp && ES_TWO_STREAMS=0 ES_WGRAD_ASYNC=0 timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/prof_ss -o p -- $CMD > /tmp/prof_ss.log 2>&1); echo "rc $?"
python tools/rocpd_stats.py "$(db ss)" $OUT/r4_single_stream_kernel_stats.txt > /dev/null
ES_DIST_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs > $OUT/r4_bench_2ranks_gloo_one_gpu.json 2> $OUT/r4_h_2ranks.err; echo "2rank rc $?"
ls -la $OUT | tail -12
