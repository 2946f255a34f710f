#!/bin/bash
# bench lines of the other BASELINE configurations (one GPU): HERA-37 fp64, HERA-350 fp32 with the "sum" regulariser,
# HERA-350 fp64, HERA-350 fp32 shared layout as the main layout
cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-cfgs}
mkdir -p $out
python3 bench.py --config hera37 --steps 500 --warmup 20 --no-cpu-baseline > $out/hera37_f64.log 2>&1
python3 bench.py --reg sum --no-cpu-baseline --no-shared > $out/hera350_f32_sum.log 2>&1
python3 bench.py --dtype f64 --no-cpu-baseline --no-shared > $out/hera350_f64.log 2>&1
python3 bench.py --layout shared --no-cpu-baseline > $out/hera350_f32_shared.log 2>&1
python3 bench.py --layout shared --reg sum --no-cpu-baseline > $out/hera350_f32_shared_sum.log 2>&1
for f in $out/*.log; do echo $f; tail -1 $f | cut -c1-330; done
