"""Aggregate rocprofv3 --pmc CSV passes (one directory per pass) into per-kernel per-dispatch averages:
   python tools/pmc_aggregate.py out.json dir1 dir2 ...     (profiles/r1/pmc_per_dispatch.json is made this way)"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, dirs = sys.argv[1], sys.argv[2:]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, set()]))
for d in dirs:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                name = row["Kernel_Name"].split("(")[0]
                cell = acc[name][row["Counter_Name"]]
                cell[0] += float(row["Counter_Value"])
                cell[1].add((path, row["Dispatch_Id"]))
res = {}
for name, counters in sorted(acc.items()):
    res[name] = {}
    for cname, (total, disp) in sorted(counters.items()):
        res[name][cname] = total / max(len(disp), 1)
        res[name]["dispatches_" + cname] = len(disp)
# the revision of the kernel sources these counters belong to: bench.py refuses a record whose blob differs from the working tree's
import hashlib

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res["_source_blob"] = {}
for src in ("kde_kernels.hip", "kde_group.hip", "stats_kernels.hip"):
    with open(os.path.join(root, "pybnesian_amd", "csrc", src), "rb") as f:
        data = f.read()
    res["_source_blob"][src] = hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()
with open(out, "w") as f:
    json.dump(res, f, indent=1)
print("kernels:", len(res))
