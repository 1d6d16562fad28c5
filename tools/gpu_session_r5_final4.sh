set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/r5_bench_default.json 2> $OUT/r5_bench_default.err; echo "bench rc $?"
bash tools/gpu_session_r5_final1.sh
