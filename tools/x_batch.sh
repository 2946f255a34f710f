#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"
bash tools/bench_round.sh r03_bench > gpurun_out/xe.log 2>&1; echo "bench_round rc $?"
out=gpurun_out/r03_bench2; mkdir -p $out
python3 bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-shared --config tutorial > $out/cfg_tutorial_f32.log 2>&1; echo "tutorial rc $?"
python3 bench.py --dist-rehearsal 8 --steps 10 --warmup 3 > $out/rehearsal8.log 2>&1; echo "rehearsal rc $?"
