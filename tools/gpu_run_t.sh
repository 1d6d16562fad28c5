#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 400 python tools/bench_loader.py --scans 8 --frames 22 --threads 1,8,32,64,128 --seconds 6 --device-workers 64 > gpurun_out/t_loader.json 2> gpurun_out/t_loader.err
grep workers gpurun_out/t_loader.err; tail -3 gpurun_out/t_loader.err | cut -c1-300; python -c "
import json; d=json.load(open('gpurun_out/t_loader.json')); print(d.get('to_device'))"
