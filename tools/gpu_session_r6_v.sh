#!/bin/bash
# round 6, session v: column sums over column groups + text-graph test; steps
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_text_graph.py tests/test_gpu_ops.py tests/test_gpu_grounding.py tests/test_gpu_config4.py tests/test_gpu_optim_table.py -m gpu -q -x > $OUT/r6v_tests.txt 2>&1; echo "rc $?"; tail -3 $OUT/r6v_tests.txt
timeout 120 python - <<'PY' | tee $OUT/r6v_colsum.txt
import torch
from embodiedscan_amd import hip
from embodiedscan_amd.hip import P, call
dev = torch.device('cuda:0'); st = torch.cuda.current_stream().cuda_stream
for n, C in ((3072, 256), (3072, 2048), (396, 256), (39048, 256), (352224, 320)):
    g = torch.randn(n, C, device=dev); dst = torch.zeros(C, device=dev)
    nws = int(hip.raw('es_colsum_workspace_floats')(n, C)); ws = torch.zeros(nws, device=dev)
    f = lambda: call('es_colsum', P(g), C, n, C, P(dst), 0, P(ws), nws, st)
    f(); f(); torch.cuda.synchronize()
    ref = g.double().sum(0)
    err = float((dst.double() - ref).abs().max() / ref.abs().max())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print(f'es_colsum {n} x {C}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us, rel err {err:.1e}')
PY
for k in grounding; do
  B="python bench.py --no-cpu-baseline --only $k --steps 10 --warmup 3 --other-steps 10"
  for rep in 1 2 3; do timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$k', d['ms_per_step'], d['value'])" | tee -a $OUT/r6v_ab.txt; done
done
B="python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 5"
for rep in 1 2; do timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mv3ddet', d['ms_per_step'], d['value'])" | tee -a $OUT/r6v_ab.txt; done
