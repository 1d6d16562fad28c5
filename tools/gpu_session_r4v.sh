#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 300 python tools/sweep_options.py --steps 12 --warmup 3 --variants "17=0;15=0;17=0,15=8192" > $OUT/r4v_sweep.txt 2> $OUT/r4v_sweep.err
cat $OUT/r4v_sweep.txt; tail -2 $OUT/r4v_sweep.err
timeout 200 python tools/bench_loader.py --scans 8 --frames 22 --threads 32,64 --kinds process --seconds 5 --device-draws > $OUT/r4v_loader_device_draws.json 2> $OUT/r4v_loader_device_draws.err; echo "rc $?"; tail -4 $OUT/r4v_loader_device_draws.err
timeout 200 python tools/bench_loader.py --scans 8 --frames 22 --threads 32,64 --kinds process --seconds 5 > $OUT/r4v_loader_exact.json 2> $OUT/r4v_loader_exact.err; echo "rc $?"; tail -4 $OUT/r4v_loader_exact.err
timeout 200 python tools/bench_loader.py --scans 8 --frames 22 --threads 32,64 --kinds process --seconds 5 --fast-draws > $OUT/r4v_loader_fast.json 2> $OUT/r4v_loader_fast.err; echo "rc $?"; tail -4 $OUT/r4v_loader_fast.err
