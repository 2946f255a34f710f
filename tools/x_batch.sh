#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"
export TMPDIR=/tmp
mkdir -p gpurun_out
V=calamity_amd/csrc/variants
timeout -k 10 600 python3 -m pytest tests/test_gpu_shapes.py tests/test_gpu_parity.py tests/test_gpu_mixed.py -q -x > gpurun_out/x10_shapes.log 2>&1; echo "shapes rc $?"; tail -4 gpurun_out/x10_shapes.log | cut -c1-300
timeout -k 10 500 python3 tools/kbench.py --layout stream --steps 20 --cache /tmp/kbs.pkl $V/lib_prev.so calamity_amd/csrc/libcalamity_hip.so > gpurun_out/x10_kb.log 2>&1 && echo kb done; cat gpurun_out/x10_kb.log
timeout -k 10 300 python3 bench.py --dist-rehearsal 8 --steps 20 --warmup 3 --no-cpu-baseline --no-shared > gpurun_out/x10_reh8.log 2>&1; echo "rehearsal rc $?"; python3 -c "
import json
l=[x for x in open('gpurun_out/x10_reh8.log') if x.startswith('{')]
d=json.loads(l[-1]); print('reh8', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['extra']['chi2_evals_per_s'])"
