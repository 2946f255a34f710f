#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_shapes.py -q -x > gpurun_out/x9_shapes.log 2>&1; echo "shapes rc $?"; tail -30 gpurun_out/x9_shapes.log | cut -c1-300
timeout -k 10 900 python3 -m pytest tests/test_gpu_config3.py -q -x > gpurun_out/x9_cfg3.log 2>&1; echo "cfg3 rc $?"; tail -5 gpurun_out/x9_cfg3.log | cut -c1-300
timeout -k 10 300 python3 bench.py --dist-rehearsal 8 --steps 20 --warmup 3 --no-cpu-baseline --no-shared > gpurun_out/x9_reh8.log 2>&1; echo "rehearsal rc $?"; tail -c 1500 gpurun_out/x9_reh8.log
