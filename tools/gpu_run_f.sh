#!/bin/bash
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_config2.py -m gpu -q -s > gpurun_out/f_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/f_pytest.log
for pp in 1 0; do
  ES_PINGPONG=$pp ONLY_L0= timeout 600 python tools/bench_conv.py 4 > gpurun_out/f_conv_pp$pp.txt 2>&1
  ES_PINGPONG=$pp timeout 600 python bench.py --no-cpu-baseline > gpurun_out/f_bench_pp$pp.json 2> gpurun_out/f_bench_pp$pp.err
  ES_PINGPONG=$pp ES_SHADOW=0 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/f_bench_pp${pp}_noshadow.json 2>> gpurun_out/f_bench_pp$pp.err
done
grep -E "passed|failed|rc=" gpurun_out/f_pytest.log
for f in gpurun_out/f_bench_pp*.json; do echo $f; head -c 330 $f | tail -c 130; echo; done
grep -E "bf16 fwd|bf16 wgrad|---" gpurun_out/f_conv_pp1.txt | head -40
echo ======; grep -E "bf16 fwd|bf16 wgrad|---" gpurun_out/f_conv_pp0.txt | head -40
