#!/usr/bin/env python
"""Print per-kernel means of the counters in a rocprofv3 --pmc csv dir: python tools/pmc_summary.py DIR [filter]"""
import collections, csv, glob, os, sys
d = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "") + " g=" + r["Grid_Size"]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    if flt in k:
        print(k, {a: round(sum(b) / len(b)) for a, b in v.items()}, "n=", len(next(iter(v.values()))))
