#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: mean counter value per dispatch of the dominant kernel.
usage: pmc_summary.py <dir> [kernel-substring] [json-out] [HMC iterations per dispatch] [second kernel, its fabric bytes beside]"""
import csv
import glob
import json
import sys
from collections import defaultdict

root = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "k_mfma32<0"
acc = defaultdict(list)
names = set()
for f in glob.glob(f"{root}/*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if kern in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
            names.add(row["Kernel_Name"].split("(")[0].replace("void ", ""))
n = max((len(v) for v in acc.values()), default=0)
print(f"kernel {kern} ({', '.join(sorted(names))}): mean per dispatch over {n} dispatches")
mean = {k: sum(v) / len(v) for k, v in acc.items()}
for k in sorted(mean):
    print(f"{k:28s} {mean[k]:18.1f}")
if "SQ_VALU_MFMA_BUSY_CYCLES" in mean and "GRBM_GUI_ACTIVE" in mean:
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles over the 1024 SIMDs
    print(f"matrix-pipe busy fraction    {mean['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / (mean['GRBM_GUI_ACTIVE'] / 8):18.3f}")
if len(sys.argv) > 3 and "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
    ipl = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    import hashlib
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for name in ("ey_mfma32.hip", "ey_common.h"):  # as bench.py's kernel_source_hash()
        h.update(open(os.path.join(root, "eeyore_amd", "csrc", name), "rb").read())
    out = {"kernel": (sorted(names)[0] if names else kern) + " (HMC trajectory)", "kernel_source_sha256": h.hexdigest(),
           "FETCH_SIZE_KB": mean["FETCH_SIZE"],
           "WRITE_SIZE_KB": mean["WRITE_SIZE"], "dispatches": n, "iterations_per_launch": ipl,
           "source": f"tools/pmc_passes.sh (rocprofv3 --pmc, separate passes), 4096 chains x L=20 x {ipl} "
                     f"iterations per dispatch"}
    if len(sys.argv) > 5:  # a kernel that runs once behind every dispatch of the first (the moments pass over the records)
        k2, acc2 = sys.argv[5], defaultdict(list)
        for f in glob.glob(f"{sys.argv[1]}/*/*/*counter_collection.csv"):
            for row in csv.DictReader(open(f)):
                if k2 in row["Kernel_Name"] and row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                    acc2[row["Counter_Name"]].append(float(row["Counter_Value"]))
        if acc2.get("FETCH_SIZE") and acc2.get("WRITE_SIZE"):
            out["behind_every_launch"] = {"kernel": k2, "FETCH_SIZE_KB": sum(acc2["FETCH_SIZE"]) / len(acc2["FETCH_SIZE"]),
                                          "WRITE_SIZE_KB": sum(acc2["WRITE_SIZE"]) / len(acc2["WRITE_SIZE"]),
                                          "dispatches": len(acc2["FETCH_SIZE"])}
            print(f"behind every launch: {k2}: FETCH_SIZE {out['behind_every_launch']['FETCH_SIZE_KB']:.1f} KB  WRITE_SIZE "
                  f"{out['behind_every_launch']['WRITE_SIZE_KB']:.1f} KB over {len(acc2['FETCH_SIZE'])} dispatches")
    json.dump(out, open(sys.argv[3], "w"), indent=1)
