#!/bin/bash
# experiment batch on the GPU box (from the repo root)
cd "${GRAFT_REPO_ROOT:?}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_launch_modes.py -q > gpurun_out/x5_modes.log 2>&1; echo "modes rc $?"; tail -12 gpurun_out/x5_modes.log | cut -c1-300
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_shapes.py tests/test_gpu_mixed.py -x -q > gpurun_out/x5_par.log 2>&1; echo "parity rc $?"; tail -5 gpurun_out/x5_par.log
