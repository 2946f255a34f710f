#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/prof_dense.sh <tag> [f32|f64]
# Counters of the dense (matrix-core) kernel on the HERA-350 shared-layout workload, each group in its own rocprofv3 pass
# (no trace domains beside --pmc): matrix-pipe busy cycles and wave state, L2 hits / misses, fabric fetch and write sizes;
# plus a kernel trace with --stats.  Everything under gpurun_out/<tag>/.
set -e
tag=${1:-dense}
dt=${2:-f32}
mops="SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16"   # (the fp32 path issues bf16 MFMAs since round 5: split_kernels.hpp)
if [ "$dt" = f64 ]; then mops=SQ_INSTS_VALU_MFMA_MOPS_F64; fi
cd "${GRAFT_REPO_ROOT:?}"
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p $out
cmd="python3 bench.py --layout shared --dtype $dt --steps 4 --warmup 1 --no-cpu-baseline --no-shared"
rocprofv3 --kernel-trace --stats -d $out/trace -o dense --output-format csv -- python3 bench.py --layout shared --dtype $dt --steps 20 --warmup 3 --no-cpu-baseline --no-shared > $out/trace_stdout.log 2>&1
echo "trace done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES $mops SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU -d $out/pmc -o sq --output-format csv -- $cmd > $out/pmc_sq.log 2>&1
echo "sq done"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $out/pmc -o tcc --output-format csv -- $cmd > $out/pmc_tcc.log 2>&1
echo "tcc done"
rocprofv3 --pmc FETCH_SIZE -d $out/pmc -o fetch --output-format csv -- $cmd > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/pmc -o write --output-format csv -- $cmd > $out/pmc_write.log 2>&1
echo "fetch/write done"
grep -h "fused_dense" $out/trace/*kernel_stats.csv | cut -c1-160
python3 tools/dense_pmc_summary.py $out $out/pmc_hera350_${dt}_shared.json
