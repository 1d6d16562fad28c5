#!/bin/bash
# round 6 session b: first GPU contact of the halo-tile kernel: its tests, the A/B tool, then the default bench + config-2 parity
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_halo.py -q -s -x > $OUT/r6b_test_halo.txt 2>&1; echo "halo tests rc $?"
tail -25 $OUT/r6b_test_halo.txt
timeout 600 python tools/bench_halo.py > $OUT/r6b_halo_ab.txt 2>&1; echo "ab rc $?"
cat $OUT/r6b_halo_ab.txt | tail -12
for h in 1 0; do
  ES_HALO=$h timeout 600 python bench.py --no-other-configs --steps 12 > $OUT/r6b_bench_halo$h.txt 2> $OUT/r6b_bench_halo$h.err; echo "bench rc $?"
  cp bench_detail.json $OUT/r6b_bench_halo${h}_detail.json
  tail -c 2100 $OUT/r6b_bench_halo$h.txt
done
timeout 900 python -m pytest tests/test_gpu_config2.py tests/test_gpu_insitu.py -q -x > $OUT/r6b_test_config2.txt 2>&1; echo "config2+insitu rc $?"
tail -5 $OUT/r6b_test_config2.txt
