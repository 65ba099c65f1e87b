#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: mean counter value per dispatch of the dominant kernel."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "k_mfma32<0>"
acc = defaultdict(list)
for f in glob.glob(f"{root}/*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if kern in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print(f"kernel {kern}: mean per dispatch over {max((len(v) for v in acc.values()), default=0)} dispatches")
for k in sorted(acc):
    v = acc[k]
    print(f"{k:28s} {sum(v) / len(v):18.1f}")
