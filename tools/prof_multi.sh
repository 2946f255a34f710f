#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/prof_multi.sh <tag> [lib.so] [f32|f64]
# Counters of the multi-slice kernels on the per-rank job of an 8-GPU run (8 time slices x 1/8 of HERA-350's baselines), each
# group in its own rocprofv3 pass (no trace domains beside --pmc), plus a kernel trace with --stats.  Under gpurun_out/<tag>/.
set -e
tag=${1:-multi}
lib=${2:-calamity_amd/csrc/libcalamity_hip.so}
dt=${3:-f32}
mops=SQ_INSTS_VALU_MFMA_MOPS_F32
if [ "$dt" = f64 ]; then mops=SQ_INSTS_VALU_MFMA_MOPS_F64; fi
cd "${GRAFT_REPO_ROOT:?}"
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p $out
cmd="python3 tools/kbench.py --child --lib $lib --dtype $dt --slices 8 --steps 4"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/trace -o multi --output-format csv -- python3 tools/kbench.py --child --lib $lib --dtype $dt --slices 8 --steps 20 > $out/trace_stdout.log 2>&1
echo "trace done"
[ -n "$SKIP_SQ" ] || timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES $mops SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU -d $out/pmc -o sq --output-format csv -- $cmd > $out/pmc_sq.log 2>&1
echo "sq done"
[ -n "$SKIP_SQ" ] || timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM -d $out/pmc -o lds --output-format csv -- $cmd > $out/pmc_lds.log 2>&1
echo "lds done"
timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $out/pmc -o tcc --output-format csv -- $cmd > $out/pmc_tcc.log 2>&1
timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $out/pmc -o ea --output-format csv -- $cmd > $out/pmc_ea.log 2>&1 || echo 'ea pass failed'
echo "tcc done"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d $out/pmc -o fetch --output-format csv -- $cmd > $out/pmc_fetch.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE -d $out/pmc -o write --output-format csv -- $cmd > $out/pmc_write.log 2>&1
timeout -k 10 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum -d $out/pmc -o tcp --output-format csv -- $cmd > $out/pmc_tcp.log 2>&1 || echo "tcp pass failed"
echo "fetch/write done"
grep -h "fused_multi\|gain_grad\|adam2" $out/trace/*kernel_stats.csv | cut -c1-200
python3 - "$out" <<'PY' | tee $out/summary.log
import csv, glob, collections, sys, os
out = sys.argv[1]
for f in sorted(glob.glob(os.path.join(out, "pmc", "*_counter_collection.csv"))):
    a = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "fused_multi" in k:
            a[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in a.items():
        print(os.path.basename(f)[:8], k[-60:], {c: round(sum(v) / len(v), 1) for c, v in d.items()})
PY
