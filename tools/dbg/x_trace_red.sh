cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/tr_red
rocprofv3 --kernel-trace --stats -d gpurun_out/tr_red -o red --output-format csv -- python3 tools/kbench.py --child --redundant --cache /tmp/kb_red.pkl --layout shared --steps 20 > gpurun_out/tr_red/stdout.log 2>&1
head -12 gpurun_out/tr_red/red_kernel_stats.csv | cut -c1-150
