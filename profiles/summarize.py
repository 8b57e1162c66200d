#!/usr/bin/env python3
"""Summarises the rocprofv3 CSVs of profiles/run_profiles.sh into a small markdown table:
per-kernel calls / total / average duration (kernel-trace --stats) and FETCH_SIZE / WRITE_SIZE per
launch (PMC passes; FETCH_SIZE x2 on gfx950 for wide coalesced reads, MI355X_MICROARCH.md §HBM)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]


def find(sub, pat):
    hits = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return hits[0] if hits else None


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = name.split("(")[0]
    return name.replace("void ", "").strip()


print(f"# rocprofv3 summary {tag}\n")
ks = find("stats", "*kernel_stats.csv")
if ks:
    print("## kernel-trace --stats (bench.py --steps 1 --warmup 1)\n")
    print("| kernel | calls | total ms | avg ms | % |")
    print("|---|---|---|---|---|")
    rows = list(csv.DictReader(open(ks)))
    ALWAYS = ("edge_pass_kernel", "edge_row_reduce_kernel", "edge_scatter_kernel", "pcent", "pair_list_kernel", "rss2_", "fit_",
              "wedge_", "bound_", "comm_", "group_eig", "group_cov", "argmax")
    for i, r in enumerate(rows):
        if i < 25 or any(a in r["Name"] for a in ALWAYS):  # the top 25 + the kernels bench.py prices against a roofline
            print(f"| {short(r['Name'])} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | "
                  f"{float(r['AverageNs'])/1e6:.4f} | {float(r['Percentage']):.2f} |")
traffic = {}
for sub, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    cc = find(sub, "*counter_collection.csv")
    if not cc:
        continue
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(cc)):
        if r.get("Counter_Name") != ctr:
            continue
        a = agg[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    print(f"\n## --pmc {ctr} (own pass)\n")
    print(f"| kernel | launches | {ctr} sum (KiB) | per launch (MB) | gfx950-corrected per launch (MB) |")
    print("|---|---|---|---|---|")
    for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        per = v * 1024 / n / 1e6
        corr = per * 2 if ctr == "FETCH_SIZE" else per
        traffic.setdefault(k, {"launches": n})[ctr + "_bytes_per_launch"] = corr * 1e6
    for i, (k, (n, v)) in enumerate(sorted(agg.items(), key=lambda kv: -kv[1][1])):
        if i >= 12 and not any(a in k for a in ("edge_pass", "edge_row", "pcent", "fit_", "wedge_", "group_eig", "group_cov")):
            continue
        per = v * 1024 / n / 1e6
        corr = per * 2 if ctr == "FETCH_SIZE" else per
        print(f"| {k} | {n} | {v:.0f} | {per:.2f} | {corr:.2f} |")
import json
for k, v in traffic.items():
    v["hbm_bytes_per_launch"] = v.get("FETCH_SIZE_bytes_per_launch", 0.0) + v.get("WRITE_SIZE_bytes_per_launch", 0.0)
with open(os.path.join(out, f"pmc_traffic_{tag}.json"), "w") as f:
    json.dump({"_provenance": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of `bench.py --steps 1 "
                              "--warmup 1`; FETCH_SIZE x2 (gfx950, wide coalesced reads), KiB -> bytes", "kernels": traffic},
              f, indent=1)
