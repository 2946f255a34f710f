#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/prof_round.sh <tag>
# rocprofv3 kernel trace + stats of the default bench, separate PMC passes (FETCH_SIZE, WRITE_SIZE) for the streaming and
# the shared layout; everything under gpurun_out/<tag>/ (tools/bench_round.sh makes the bench lines once the summaries are in profiles/)
set -e
tag=${1:-prof}
cd "${GRAFT_REPO_ROOT:?}"
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/trace -o bench --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $out/trace_stdout.log 2>&1
echo "trace done"
rocprofv3 --pmc FETCH_SIZE -d $out/pmc -o fetch --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-shared > $out/pmc_fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE -d $out/pmc -o write --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-shared > $out/pmc_write.log 2>&1
echo "write done"
python3 tools/pmc_summary.py $out/pmc/fetch_counter_collection.csv $out/pmc/write_counter_collection.csv $out/pmc_hera350_f32_stream.json > $out/pmc_summary.log 2>&1
tail -2 $out/pmc_summary.log | cut -c1-300
