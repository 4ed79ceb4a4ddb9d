#!/usr/bin/env python3
"""rocprofv3 --pmc passes of tools/pmc_scene.sh -> one entry of profiles/hbm_traffic.json and a text summary.

    tools/pmc_to_json.py gpurun_out/pmc_<tag> <kernel substring> <json key> [summary.txt]

HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (both reported in KiB; gfx950 tallies a 128-byte
read request as 64 bytes, so FETCH_SIZE is doubled; WRITE_SIZE as reported), each from a pass of
its own, per dispatch of the named kernel."""
import collections
import csv
import glob
import json
import os
import sys

root, sub, key = sys.argv[1], sys.argv[2], sys.argv[3]
tot = collections.defaultdict(float)
disp = collections.defaultdict(set)
def newest_per_directory(pattern):
    """One file per directory, the newest: a directory merged from several collections of the same tag (gpurun_out is
    merged into, never cleared) would otherwise average the counters of different builds."""
    best = {}
    for f in glob.glob(pattern, recursive=True):
        d = os.path.dirname(f)
        if d not in best or os.path.getmtime(f) > os.path.getmtime(best[d]):
            best[d] = f
    return sorted(best.values())


for f in newest_per_directory(os.path.join(root, "**", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            disp[r["Counter_Name"]].add((f, r["Dispatch_Id"]))
per = {k: tot[k] / len(disp[k]) for k in tot}
if not per.get("SQ_INSTS_VALU") or "FETCH_SIZE" not in per:
    sys.exit("no counters of a kernel matching %r under %s: nothing written" % (sub, root))
lines = ["%-30s %18.0f  per-dispatch %16.0f  (%d dispatches)" % (k, tot[k], per[k], len(disp[k])) for k in sorted(tot)]
# kernel durations from the trace pass
for f in newest_per_directory(os.path.join(root, "trace", "**", "*kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        lines.append("kernel_stats: %s calls %s avg_ns %s pct %s" % (r["Name"][:60], r["Calls"], r["AverageNs"], r["Percentage"]))
entry = {
    "hbm_bytes_per_launch": int(2 * per.get("FETCH_SIZE", 0) * 1024 + per.get("WRITE_SIZE", 0) * 1024),
    "fetch_size_kb_per_launch_raw": int(per.get("FETCH_SIZE", 0)), "write_size_kb_per_launch_raw": int(per.get("WRITE_SIZE", 0)),
    "correction": "FETCH_SIZE doubled (gfx950 tallies 128-byte read requests at 64 bytes); WRITE_SIZE as reported",
    "kernel": sub,
}
if per.get("SQ_INSTS_VALU"):
    entry["valu_active_lane_fraction"] = round(per["SQ_THREAD_CYCLES_VALU"] / (64.0 * per["SQ_INSTS_VALU"]), 4)
    entry["valu_wave_instructions_per_launch"] = int(per["SQ_INSTS_VALU"])
if per.get("TCC_REQ_sum"):
    entry["l2_hit_rate"] = round(per["TCC_HIT_sum"] / max(1.0, per["TCC_HIT_sum"] + per["TCC_MISS_sum"]), 4)
if per.get("SQ_WAVE_CYCLES") and per.get("SQ_WAIT_ANY"):
    entry["wave_cycles_waiting_fraction"] = round(per["SQ_WAIT_ANY"] / per["SQ_WAVE_CYCLES"], 4)
if per.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
    entry["l1_accesses_per_launch"] = int(per["TCP_TOTAL_CACHE_ACCESSES_sum"])
    entry["l1_to_l2_read_requests_per_launch"] = int(per.get("TCP_TCC_READ_REQ_sum", 0))
if per.get("TA_TA_BUSY_sum"):
    entry["ta_busy_cycles_per_launch"] = int(per["TA_TA_BUSY_sum"])
summary = sys.argv[4] if len(sys.argv) > 4 else None
entry["source"] = "%s: rocprofv3 --pmc passes (one counter group per process) over `bench.py --steps 2 --warmup 1`, per dispatch of %s" % (
    summary or root, sub)
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "hbm_traffic.json")
try:
    allj = json.load(open(path))
except Exception:
    allj = {}
allj[key] = entry
json.dump(allj, open(path, "w"), indent=1)
text = "\n".join(lines)
if summary:
    open(summary, "w").write("# %s  kernel %s\n" % (key, sub) + text + "\n")
print(text)
print(json.dumps(entry, indent=1))
