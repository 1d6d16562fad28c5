#!/bin/bash
# round 6, session q: stem kernel without SGPR spills, LayerNorm backward with two rows in flight: tests, timings, grounding / mv-3ddet / occupancy steps
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_imgconv.py tests/test_gpu_grounding.py tests/test_gpu_config4.py tests/test_gpu_optim_table.py tests/test_gpu_resnet2d.py -m gpu -q -s -x > $OUT/r6q_tests.txt 2>&1; echo "rc $?"; grep 'stem + pool\|passed\|failed' $OUT/r6q_tests.txt | cut -c1-200
timeout 120 python - <<'PY' | tee $OUT/r6q_ln_bwd.txt
import torch
from embodiedscan_amd import hip
from embodiedscan_amd.hip import P, call
dev = torch.device('cuda:0'); st = torch.cuda.current_stream().cuda_stream
for n, C in ((3072, 256), (396, 256), (39048, 256)):
    dy, z = torch.randn(n, C, device=dev), torch.randn(n, C, device=dev)
    w = torch.rand(C, device=dev) + .5
    mean, rstd = torch.randn(n, device=dev), torch.rand(n, device=dev) + .5
    dz, dw, db = torch.empty(n, C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    nws = int(hip.raw('es_layernorm_bwd_workspace_floats')(n, C)); ws = torch.zeros(nws, device=dev)
    f = lambda: call('es_layernorm_bwd', P(dy), P(z), n, C, P(w), P(mean), P(rstd), P(dz), 0, P(dw), P(db), P(ws), nws, st)
    f(); f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print(f'es_layernorm_bwd {n} x {C}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us')
PY
for k in grounding occupancy; do
  B="python bench.py --no-cpu-baseline --only $k --steps 10 --warmup 3 --other-steps 10"
  for rep in 1 2; do timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$k', d['ms_per_step'], d['value'])" | tee -a $OUT/r6q_ab.txt; done
done
B="python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 5"
for rep in 1 2; do timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mv3ddet', d['ms_per_step'], d['value'])" | tee -a $OUT/r6q_ab.txt; done
