#!/bin/bash
# experiment batch on the GPU box (from the repo root): kernel timings of library variants + diagnostics
cd "${GRAFT_REPO_ROOT:?}"
export TMPDIR=/tmp
mkdir -p gpurun_out
V=calamity_amd/csrc/variants
timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_shapes.py -m gpu -x -q > gpurun_out/x3_pytest.log 2>&1; echo "pytest rc $?"
tail -3 gpurun_out/x3_pytest.log
timeout -k 10 400 python3 tools/kbench.py --layout shared --steps 20 $V/lib_r2.so calamity_amd/csrc/libcalamity_hip.so > gpurun_out/x3_kb.log 2>&1 && echo kb done
timeout -k 10 300 python3 tools/kbench.py --layout shared --dtype f64 --steps 20 --cache /tmp/kb64.pkl $V/lib_r2.so calamity_amd/csrc/libcalamity_hip.so > gpurun_out/x3_kb64.log 2>&1 && echo kb64 done
cat gpurun_out/x3_kb.log gpurun_out/x3_kb64.log
timeout -k 10 600 bash tools/prof_dense.sh x3_dense f32 > gpurun_out/x3_prof.log 2>&1; echo "prof rc $?"; tail -5 gpurun_out/x3_prof.log
timeout -k 10 500 python3 tools/dropin_breakdown.py --maxsteps 1000 > gpurun_out/x3_dropin.log 2>&1; echo "dropin rc $?"; tail -c 1500 gpurun_out/x3_dropin.log
