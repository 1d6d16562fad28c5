#!/bin/bash
# round-3 session C: full GPU suite, default bench line, f64 calibration of the config-5 gradients, kernel trace of mv-3ddet
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/r3_c_pytest.txt 2>&1
echo "pytest rc $?" >> gpurun_out/r3_c_pytest.txt
tail -12 gpurun_out/r3_c_pytest.txt
timeout 900 python bench.py > gpurun_out/r3_c_bench.json 2> gpurun_out/r3_c_bench.err
echo "bench rc $?"
tail -c 600 gpurun_out/r3_c_bench.err
head -c 300 gpurun_out/r3_c_bench.json
timeout 600 python tools/calib_config5.py > gpurun_out/r3_config5_f64_calibration.txt 2> gpurun_out/r3_c_calib.err
echo "calib rc $?"; tail -c 300 gpurun_out/r3_c_calib.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c -o p -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-other-configs --steps 4 --warmup 2 > /tmp/prof_c.log 2>&1
echo "rocprof rc $?"
cd "$GRAFT_REPO_ROOT"
DB=$(find /tmp/prof_c -name '*.db' | head -1)
python tools/rocpd_stats.py "$DB" gpurun_out/r3_c_kernel_stats.txt > /dev/null 2>&1 || (tail -20 /tmp/prof_c.log; ls -R /tmp/prof_c | head)
head -30 gpurun_out/r3_c_kernel_stats.txt
