#!/bin/bash
# usage: gpu_step.sh <tag> <libs...>   -- parity tests on the default build, then kbench f32 (+ f64 if KB64=1) on the given libs
set -e
cd $GRAFT_REPO_ROOT
tag=$1; shift
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/t_$tag.log 2>&1 || { tail -30 gpurun_out/t_$tag.log; exit 1; }
tail -2 gpurun_out/t_$tag.log
timeout -k 10 500 python tools/kbench.py "$@" > gpurun_out/kb_$tag.log 2>&1
if [ -n "$KB64" ]; then timeout -k 10 500 python tools/kbench.py --dtype f64 --cache /tmp/kb64.pkl "$@" >> gpurun_out/kb_$tag.log 2>&1; fi
cut -c1-170 gpurun_out/kb_$tag.log
