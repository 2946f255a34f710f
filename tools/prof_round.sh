set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_r1c gpurun_out/pmc_r1c
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r1c -o bench --output-format csv -- python3 bench.py --steps 20 --warmup 3 > gpurun_out/prof_r1c/stdout.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_r1c -o fetch --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-shared --no-cpu-baseline > gpurun_out/pmc_r1c/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_r1c -o write --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-shared --no-cpu-baseline > gpurun_out/pmc_r1c/write.log 2>&1
python3 bench.py --steps 20 --warmup 3 > gpurun_out/bench_r1c.log 2>&1
tail -1 gpurun_out/bench_r1c.log
