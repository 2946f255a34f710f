#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"
timeout -k 10 400 python3 -m pytest tests/test_gpu_shapes.py -q -x -k "share" > gpurun_out/x28.log 2>&1; rc=$?; echo "rc $rc"; tail -4 gpurun_out/x28.log | cut -c1-250
for n in 2 4 8; do timeout -k 10 300 python3 tools/kbench.py --child --slices $n --steps 20 2>&1 | grep KBENCH | sed "s/^/slices $n /" >> gpurun_out/x28_k.log; done; cat gpurun_out/x28_k.log
