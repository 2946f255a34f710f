#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) into per-launch HBM bytes per kernel.

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): counters are in KiB; on gfx950 FETCH_SIZE
reports exactly half of the bytes of a wide coalesced streaming read, so it is doubled; WRITE_SIZE is exact for streaming stores.
usage: pmc_summary.py fetch_counter_collection.csv write_counter_collection.csv out.json
"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v)) for k, v in agg.items()}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    import bench

    out = {"kernel_source_hash": bench.kernel_source_hash(), "units": "bytes per launch", "fetch_correction": "FETCH_SIZE KiB x 1024 x 2 (gfx950 wide-read undercount)", "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("void calk::") and "calk::" not in k:
            continue
        f = fetch.get(k, (0, 0.0))
        w = write.get(k, (0, 0.0))
        out["kernels"][k.split("(")[0]] = {
            "launches_profiled": max(f[0], w[0]),
            "fetch_bytes": f[1] * 1024 * 2,
            "write_bytes": w[1] * 1024,
            "hbm_bytes": f[1] * 1024 * 2 + w[1] * 1024,
        }
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in out["kernels"].items():
        print(f"{k[:70]:70s} fetch {v['fetch_bytes'] / 1e9:8.3f} GB  write {v['write_bytes'] / 1e9:7.3f} GB")


if __name__ == "__main__":
    main()
