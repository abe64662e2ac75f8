#!/usr/bin/env python3
"""Reduce rocprofv3 CSV output (kernel stats + PMC counter collections) to the kws_* kernels and print/save a
compact per-kernel summary.  Usage: tools/summarize_prof.py <gpurun_out dir> <tag> [profiles/<round>] -> <tag>_kernel_stats.csv, <tag>_summary.json"""
import collections
import csv
import glob
import hashlib
import json
import os
import sys


def newest(src, pattern):
    """gpurun MERGES what a call wrote into the local gpurun_out/: a directory that several calls profiled into holds one file set per call (named by
    process id).  Only the newest file of each directory belongs to the run being summarised."""
    by_dir = {}
    for path in glob.glob(os.path.join(src, "**", pattern), recursive=True):
        d = os.path.dirname(path)
        if d not in by_dir or os.path.getmtime(path) > os.path.getmtime(by_dir[d]):
            by_dir[d] = path
    return sorted(by_dir.values())


def main(src, tag, dst="profiles/r04"):
    os.makedirs(dst, exist_ok=True)
    summary = {}
    for path in newest(src, "*_kernel_stats.csv"):
        rows = [r for r in csv.DictReader(open(path)) if "kws::" in r["Name"] or "nccl" in r["Name"].lower()]
        with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows)
        for r in rows:
            name = r["Name"].split("(")[0].replace("void ", "")
            summary.setdefault(name, {})["avg_ms"] = float(r["AverageNs"]) / 1e6
            summary[name]["calls"] = int(r["Calls"])
    for path in newest(src, "*_counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        meta = {}
        for r in csv.DictReader(open(path)):
            if "kws::" not in r["Kernel_Name"]:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = {"vgpr": int(r["VGPR_Count"]), "agpr": int(r["Accum_VGPR_Count"]), "sgpr": int(r["SGPR_Count"]),
                       "lds": int(r["LDS_Block_Size"]), "scratch": int(r["Scratch_Size"]), "grid": int(r["Grid_Size"])}
        for k, d in agg.items():
            s = summary.setdefault(k, {})
            s.update(meta[k])
            for c, v in d.items():
                s[c] = sum(v) / len(v)
    # digests of the kernel sources as profiled (run this right after the profile, before editing them): bench.py compares
    # the digest with the tree's before it quotes the counters of this summary
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    summary["_sources"] = {}
    for path in sorted(glob.glob(os.path.join(root, "honk2_amd", "csrc", "*.hip"))):
        with open(path, "rb") as f:
            summary["_sources"][os.path.relpath(path, root)] = hashlib.sha256(f.read()).hexdigest()
    with open(os.path.join(dst, f"{tag}_summary.json"), "w") as f:
        json.dump(summary, f, indent=1, sort_keys=True)
    print(json.dumps(summary, indent=1, sort_keys=True))


if __name__ == "__main__":
    main(*sys.argv[1:])
