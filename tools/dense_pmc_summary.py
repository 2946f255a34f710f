#!/usr/bin/env python3
"""Summarise the rocprofv3 passes of tools/prof_dense.sh for the dense (matrix-core) kernel: matrix-pipe busy share, wave
states, L2 hit rate, fabric fetch / write bytes per launch.

Units follow /opt/skills/guides/MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD (64 per
v_mfma_f32_32x32x2_f32), SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are wave-state counters (ratios are used only);
FETCH_SIZE / WRITE_SIZE are KiB, FETCH_SIZE doubled (gfx950 wide-read undercount).
usage: dense_pmc_summary.py <gpurun_out/tag> out.json
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N_SIMD = 256 * 4


def agg(path):
    a = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "fused_dense" in k:
            a[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in a.items()}


def main():
    src, out_path = sys.argv[1], sys.argv[2]
    import bench

    stats = {}
    for r in csv.DictReader(open(glob.glob(os.path.join(src, "trace", "*kernel_stats.csv"))[0])):
        if "fused_dense" in r["Name"]:
            stats[r["Name"].split("(")[0].replace("void ", "")] = dict(calls=int(r["Calls"]), avg_ms=float(r["AverageNs"]) / 1e6)
    pm = {f: agg(glob.glob(os.path.join(src, "pmc", f + "_counter_collection.csv"))[0]) for f in ("sq", "tcc", "fetch", "write")}
    out = {"kernel_source_hash": bench.kernel_source_hash(), "note": "per launch; durations from the un-instrumented --kernel-trace --stats run, counters from separate --pmc passes",
           "kernels": {}}
    for k, st in sorted(stats.items()):
        sq, tcc = pm["sq"].get(k, {}), pm["tcc"].get(k, {})
        e = dict(st)
        if sq:
            # 512-flop units: a v_mfma_f32_32x32x2_f32 is 4096 flops (64 pipe cycles), a v_mfma_f64_16x16x4_f64 2048 (64 cycles)
            if sq.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0) > 0:  # the split-bf16 kernel: a v_mfma_f32_32x32x16_bf16 is 32768 flops = 64 units (32 pipe cycles)
                n_mfma = sq["SQ_INSTS_VALU_MFMA_MOPS_BF16"] / 64.0
            else:
                n_mfma = sq["SQ_INSTS_VALU_MFMA_MOPS_F32"] / 8.0 if sq.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0) > 0 else sq["SQ_INSTS_VALU_MFMA_MOPS_F64"] / 4.0
            e.update(mfma_instructions=n_mfma, mfma_busy_cycles=sq["SQ_VALU_MFMA_BUSY_CYCLES"],
                     mfma_busy_cycles_per_instruction=sq["SQ_VALU_MFMA_BUSY_CYCLES"] / max(n_mfma, 1),
                     wave_wait_share=sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"], wave_issue_stall_share=sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"],
                     wave_active_share=sq["SQ_ACTIVE_INST_ANY"] / sq["SQ_WAVE_CYCLES"])
            # share of the SIMD-cycles of the launch (at the clock the matrix pipe needs to fit its busy cycles into the
            # measured duration the share would be 1): busy cycles / (duration x 2.4 GHz x 1024 SIMDs) is a LOWER bound of the
            # pipe utilisation, the chip runs below 2.4 GHz under this load
            e["mfma_busy_share_at_2p4GHz"] = sq["SQ_VALU_MFMA_BUSY_CYCLES"] / (st["avg_ms"] * 1e-3 * 2.4e9 * N_SIMD)
        if tcc:
            e.update(l2_requests=tcc["TCC_REQ_sum"], l2_hit_rate=tcc["TCC_HIT_sum"] / max(tcc["TCC_HIT_sum"] + tcc["TCC_MISS_sum"], 1))
        f, w = pm["fetch"].get(k, {}).get("FETCH_SIZE"), pm["write"].get(k, {}).get("WRITE_SIZE")
        if f is not None:
            e["fetch_bytes"] = f * 1024 * 2
        if w is not None:
            e["write_bytes"] = w * 1024
        out["kernels"][k] = e
    g = [v for k, v in out["kernels"].items() if "<true" in k]
    if g:
        out["gradient_pass"] = dict(ms=sum(v["avg_ms"] for v in g), hbm_bytes=sum(v.get("fetch_bytes", 0) + v.get("write_bytes", 0) for v in g),
                                    mfma_busy_share_at_2p4GHz=sum(v.get("mfma_busy_cycles", 0) for v in g) / (sum(v["avg_ms"] for v in g) * 1e-3 * 2.4e9 * N_SIMD))
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out.get("gradient_pass")))


if __name__ == "__main__":
    main()
