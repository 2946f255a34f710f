#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/bench_round.sh <tag>
# The bench lines of a round, after the counter summaries of tools/prof_round.sh / tools/prof_dense.sh have been copied
# into profiles/ (bench.py reports roofline.traffic from them when they match the kernel sources): the default run with
# the CPU baseline, the extras, and the other BASELINE configurations.  Everything under gpurun_out/<tag>/.
set -e
tag=${1:-bench}
cd "${GRAFT_REPO_ROOT:?}"
out=gpurun_out/$tag
mkdir -p $out
python3 bench.py --steps 20 --warmup 3 > $out/bench.log 2>&1
echo "default done"
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --extras > $out/bench_extras.log 2>&1
echo "extras done"
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-shared --layout shared > $out/cfg_hera350_f32_shared.log 2>&1
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-shared --layout shared --dtype f64 > $out/cfg_hera350_f64_shared.log 2>&1
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-shared --dtype f64 > $out/cfg_hera350_f64.log 2>&1
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-shared --reg sum > $out/cfg_hera350_f32_sum.log 2>&1
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-shared --reg sum --layout shared > $out/cfg_hera350_f32_shared_sum.log 2>&1
python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-shared --config hera37 --dtype f64 > $out/cfg_hera37_f64.log 2>&1
echo "configs done"
tail -1 $out/bench.log | cut -c1-600
