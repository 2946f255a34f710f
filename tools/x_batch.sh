#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"
export TMPDIR=/tmp
mkdir -p gpurun_out
V=calamity_amd/csrc/variants
timeout -k 10 400 python3 tools/kbench.py --layout shared --dtype f64 --steps 20 --cache /tmp/kb64.pkl calamity_amd/csrc/libcalamity_hip.so $V/lib_norelax64.so > gpurun_out/x12_kb64.log 2>&1 && echo kb64 done; cat gpurun_out/x12_kb64.log
timeout -k 10 500 python3 tools/dropin_breakdown.py --maxsteps 1000 > gpurun_out/x12_dropin.log 2>&1; echo "dropin rc $?"
python3 -c "
import json
l=[x for x in open('gpurun_out/x12_dropin.log') if x.startswith('{')]
d=json.loads(l[-1]); print(d['total_s'], d['other_s']); [print('   %.3f %s'%(v,k)) for k,v in d['split_s'].items()]"
