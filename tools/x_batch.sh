#!/bin/bash
# the round's profiles (tools/prof_round.sh, tools/prof_dense.sh) in one call
cd "${GRAFT_REPO_ROOT:?}"
bash tools/prof_round.sh r03_final && bash tools/prof_dense.sh r03_dense_f32 f32 && bash tools/prof_dense.sh r03_dense_f64 f64
