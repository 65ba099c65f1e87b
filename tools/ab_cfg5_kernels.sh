#!/bin/bash
# Per-kernel times (rocprofv3 kernel trace) and LDS bank-conflict counters of config 5's share for several library builds.
# usage (on the GPU box, repo root): tools/ab_cfg5_kernels.sh libA libB ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  tag=$(basename $lib .so); out=gpurun_out/abk_$tag; rm -rf $out; mkdir -p $out
  export EEYORE_AMD_LIB=$PWD/$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/bench_config5.py ${AB_CHAINS:-4096} 1 > $out/run.log 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $out/b -- python3 tools/bench_config5.py ${AB_CHAINS:-4096} 1 > /dev/null 2>&1
  echo "== $lib: $(grep -h 'leapfrog' $out/run.log | tail -1 | cut -c1-150)"
  python3 - "$out" <<'PY'
import csv, glob, sys, re
from collections import defaultdict
root = sys.argv[1]
f = glob.glob(f"{root}/trace/*/*kernel_stats.csv")[0]
for row in list(csv.DictReader(open(f)))[:6]:
    print(f"   {row['Name'][:60]:60s} calls {row['Calls']:>5s} avg {float(row['AverageNs'])/1e6:8.3f} ms  {row['Percentage']} %")
tot = defaultdict(lambda: defaultdict(float))
for f in glob.glob(f"{root}/b/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        tot[re.sub(r"\(.*", "", r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in tot.items():
    if v.get("SQ_LDS_IDX_ACTIVE", 0) > 1e9:
        print(f"   LDS conflicts / active: {v['SQ_LDS_BANK_CONFLICT'] / v['SQ_LDS_IDX_ACTIVE']:.3f}   {k[:70]}")
PY
done
