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
python3 bench.py --steps 2000 --warmup 100 --no-cpu-baseline --no-shared --config tutorial > $out/cfg_tutorial_f32.log 2>&1
# the whole 8-slice job of BASELINE config 3 on ONE GPU (the multi-rank code path with one rank): 8 slices x 61 075 baselines, tiles shared
python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-shared --dist-rehearsal 8 > $out/cfg_hera350_f32_rehearsal8.log 2>&1
python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-shared --dist-rehearsal 8 --reg sum > $out/cfg_hera350_f32_rehearsal8_sum.log 2>&1
python3 tools/slices_bench.py --T 1 4 16 32 64 > $out/slices_tutorial.log 2>&1
echo "configs done"
tail -1 $out/bench.log | cut -c1-600
