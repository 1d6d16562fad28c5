#!/bin/bash
# round 6 session a: the default bench line in its new short form (+ detail file) and the per-launch dump of the single-stream step
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
ES_BENCH_DUMP=$OUT/r6a_launches_mv3ddet.jsonl timeout 900 python bench.py > $OUT/r6a_bench_stdout.txt 2> $OUT/r6a_bench.err; echo "bench rc $?"
tail -c 4200 $OUT/r6a_bench_stdout.txt
wc -c $OUT/r6a_bench_stdout.txt
cp bench_detail.json $OUT/r6a_bench_detail.json
