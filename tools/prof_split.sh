#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/prof_split.sh <tag>
# Counters of the split-bf16 dense kernel (fused_dense_split2_kernel) on the HERA-350 shared-layout workload, each group in its own
# rocprofv3 pass (no trace domains beside --pmc), plus a kernel trace with --stats.  Everything under gpurun_out/<tag>/.
tag=${1:-split}
cd "${GRAFT_REPO_ROOT:?}"
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p $out
cmd="python3 bench.py --layout shared --dtype f32 --steps 4 --warmup 1 --no-cpu-baseline --no-shared"
rocprofv3 --kernel-trace --stats -d $out/trace -o dense --output-format csv -- python3 bench.py --layout shared --dtype f32 --steps 20 --warmup 3 --no-cpu-baseline --no-shared > $out/trace_stdout.log 2>&1
echo "trace done"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $out/pmc -o p$i --output-format csv -- $cmd > $out/pmc_p$i.log 2>&1
  echo "pass $i rc=$?"
done
grep -h "fused_dense\|gain_grad\|adam2" $out/trace/*kernel_stats.csv | cut -c1-200
python3 - <<PY
import csv, glob, collections
a = collections.defaultdict(list)
for f in sorted(glob.glob("$out/pmc/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "fused_dense_split2_kernel<true>" in r["Kernel_Name"].replace(" ", "") or "fused_dense_split2_kernelILb1" in r["Kernel_Name"]:
            a[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in a.items():
    print(f"{k:34s} {sum(v) / len(v):16.0f}  ({len(v)} launches)")
PY
