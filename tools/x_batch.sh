#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"
out=gpurun_out/r03_bench2; mkdir -p $out
python3 bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-shared --config tutorial > $out/cfg_tutorial_f32.log 2>&1; echo "tutorial rc $?"
python3 tools/dropin_breakdown.py --reg sum > $out/dropin_hera350_sum.log 2>&1; echo "dropin sum rc $?"; tail -c 600 $out/dropin_hera350_sum.log
python3 tools/dropin_breakdown.py --reg post_hoc > $out/dropin_hera350_post_hoc.log 2>&1; echo "dropin post_hoc rc $?"
python3 bench.py --dist-rehearsal 8 --steps 10 --warmup 3 > $out/rehearsal8.log 2>&1; echo "rehearsal rc $?"; tail -1 $out/rehearsal8.log | cut -c1-300
