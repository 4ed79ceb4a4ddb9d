#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter CSVs per kernel:  tools/pmc_summary.py gpurun_out/pmc_<tag> [kernel-substring]"""
import csv, glob, os, sys, collections
root = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else "k_render_mega"
tot = collections.defaultdict(float); disp = collections.defaultdict(set)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); disp[r["Counter_Name"]].add((f, r["Dispatch_Id"]))
for k in sorted(tot):
    n = len(disp[k]); print("%-28s %18.0f  per-dispatch %16.0f  (%d dispatches)" % (k, tot[k], tot[k]/n, n))
