#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"
timeout -k 10 900 python3 -m pytest tests/test_gpu_shapes.py tests/test_gpu_config3.py tests/test_gpu_distributed.py tests/test_gpu_exchange_hook.py -q -x > gpurun_out/xt.log 2>&1; rc=$?; echo "tests rc $rc"; tail -3 gpurun_out/xt.log | cut -c1-250
