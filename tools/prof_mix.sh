#!/bin/bash
# usage (GPU box, repo root): bash tools/prof_mix.sh <tag>  -- instruction mix and wave-state counters of the dense kernel
set -e
tag=${1:-mix}
cd "${GRAFT_REPO_ROOT:?}"
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p $out
cmd="python3 bench.py --layout shared --steps 4 --warmup 1 --no-cpu-baseline --no-shared"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD -d $out/pmc -o mix1 --output-format csv -- $cmd > $out/mix1.log 2>&1
echo "mix1 done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU -d $out/pmc -o mix2 --output-format csv -- $cmd > $out/mix2.log 2>&1
echo "mix2 done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_IFETCH SQC_ICACHE_MISSES SQC_ICACHE_REQ SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES -d $out/pmc -o mix3 --output-format csv -- $cmd > $out/mix3.log 2>&1
echo "mix3 done"
TAG=$tag python3 - <<'PY'
import csv, glob, collections, os
out = os.environ.get("OUTDIR", "")
for f in sorted(glob.glob("gpurun_out/%s/pmc/mix*_counter_collection.csv" % os.environ.get("TAG", "mix"))):
    a = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "fused_dense" in k and "<true" in k.replace("(bool)1", "true"):
            a[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in a.items():
        print(os.path.basename(f), k[-40:], {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
