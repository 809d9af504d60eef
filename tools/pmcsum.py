#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per kernel."""
import csv
import glob
import sys
from collections import defaultdict

for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = defaultdict(list)
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("shk::", "")
            acc[(k, row["Counter_Name"])].append(float(row["Counter_Value"]))
        for (k, c), v in sorted(acc.items()):
            if k.startswith("k_"):
                print(f"{k:18s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
