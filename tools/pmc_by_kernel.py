#!/usr/bin/env python3
"""rocprofv3 --pmc CSV output summarised per kernel name: dispatches and the SUM of every counter over them, plus the
matrix-pipe busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs) where both were collected.
usage: pmc_by_kernel.py <dir>"""
import csv
import glob
import re
import sys
from collections import defaultdict

root = sys.argv[1]
tot = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(f"{root}/*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", row["Kernel_Name"])
        name = re.sub(r"^void ", "", name)
        tot[name][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[name][row["Counter_Name"]] += 1
for name in sorted(tot, key=lambda k: -tot[k].get("SQ_BUSY_CYCLES", tot[k].get("GRBM_GUI_ACTIVE", 0.0))):
    c = tot[name]
    n = max(cnt[name].values())
    line = f"{name[:70]:70s} dispatches {n:5d}"
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"] > 0:
        line += f"  mfma_busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / (c['GRBM_GUI_ACTIVE'] / 8):.3f}"
    print(line)
    for k in sorted(c):
        print(f"    {k:28s} {c[k]:20.1f}")
