"""Summarise rocprofv3 output directories: kernel-trace stats (avg duration per kernel) and --pmc counter passes
(mean counter value per launch of each kernel).  usage: pmc_summary.py DIR [DIR ...] > summary.json"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = {}
# which build of the library the profiled command ran on: the stamp next to libcopra_hip.so (= copra_source_hash() of that library);
# bench.py flags counters taken on another build as stale
_stamp = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "copra_amd", "csrc", "libcopra_hip.so.srchash")
out["library_source_hash"] = open(_stamp).read().strip() if os.path.exists(_stamp) else None
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            out.setdefault("kernel_stats", []).append({k: row[k] for k in ("Name", "Calls", "TotalDurationNs",
                                                                             "AverageNs", "Percentage") if k in row})
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(list)
        for row in csv.DictReader(open(f)):
            acc[(row["Kernel_Name"], row["Counter_Name"])].append(float(row["Counter_Value"]))
        for (kern, ctr), vals in acc.items():
            out.setdefault("counters", {}).setdefault(kern, {})[ctr] = {"launches": len(vals),
                                                                         "mean_per_launch": sum(vals) / len(vals)}
json.dump(out, sys.stdout, indent=1)
