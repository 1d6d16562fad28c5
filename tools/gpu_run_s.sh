#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 1200 python tools/bench_loader.py --scans 8 --frames 22 --threads 1,8,16,32,64,128 --repeat 8 > gpurun_out/s_loader.json 2> gpurun_out/s_loader.err
grep workers gpurun_out/s_loader.err; tail -3 gpurun_out/s_loader.err | cut -c1-300; python -c "
import json; d=json.load(open('gpurun_out/s_loader.json')); print(d.get('to_device'))"
