#!/bin/bash
# round 5, session e: collector frozen after warm-up (step-time outliers), from-files leg by worker count, host profile of the
# grounding step
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
ES_BENCH_DIAG=1 timeout 500 python bench.py --no-cpu-baseline --only grounding --steps 40 --other-steps 40 --warmup 3 > $OUT/r5e_bench_grounding_diag.json 2> $OUT/r5e_bench_grounding_diag.err; echo "rc $?"
for w in 10 12 14; do
  ES_LOADER_WORKERS=$w timeout 300 python bench.py --no-cpu-baseline --only from_files --steps 24 --other-steps 24 > $OUT/r5e_bench_from_files_w$w.json 2> $OUT/r5e_bench_from_files_w$w.err; echo "rc $?"
done
timeout 300 python bench.py --no-cpu-baseline --no-other-configs --steps 30 --warmup 5 > $OUT/r5e_bench_mv3ddet.json 2> $OUT/r5e_bench_mv3ddet.err; echo "rc $?"
timeout 400 python -m cProfile -o /tmp/ground.prof bench.py --no-cpu-baseline --only grounding --steps 12 --other-steps 12 --warmup 3 > /dev/null 2> $OUT/r5e_cprofile.err; echo "rc $?"
python - > $OUT/r5e_cprofile_grounding.txt 2>&1 <<'PY'
import pstats
p = pstats.Stats('/tmp/ground.prof')
p.sort_stats('tottime').print_stats(45)
p.sort_stats('cumulative').print_stats(60)
PY
