#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"
bash tools/prof_round.sh r03_final > gpurun_out/xa.log 2>&1; echo "prof_round rc $?"
bash tools/prof_dense.sh r03_dense_f32 f32 > gpurun_out/xb.log 2>&1; echo "prof_dense f32 rc $?"; tail -1 gpurun_out/xb.log | cut -c1-300
bash tools/prof_dense.sh r03_dense_f64 f64 > gpurun_out/xc.log 2>&1; echo "prof_dense f64 rc $?"; tail -1 gpurun_out/xc.log | cut -c1-300
bash tools/prof_multi.sh r03_multi > gpurun_out/xd.log 2>&1; echo "prof_multi rc $?"
