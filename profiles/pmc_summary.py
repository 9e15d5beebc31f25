#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel."""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows:
            k = r["Kernel_Name"]
            if "mj_k" not in k:
                continue
            k = k[k.index("mj_k"):].split("(")[0]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("==", f)
        for k, cs in agg.items():
            print(" ", k, {c: round(sum(v) / len(v)) for c, v in cs.items()}, "n=", len(next(iter(cs.values()))))
