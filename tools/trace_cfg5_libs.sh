#!/bin/bash
# Per-kernel times (rocprofv3 kernel trace) of config 5's share for several library builds.  usage: tools/trace_cfg5_libs.sh libA libB ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  tag=$(basename $lib .so); out=gpurun_out/trl_$tag; rm -rf $out; mkdir -p $out
  EEYORE_AMD_LIB=$PWD/$lib rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/bench_config5.py ${AB_CHAINS:-4096} 1 > $out/run.log 2>&1
  echo "== $lib: $(grep -h 'leapfrog' $out/run.log | tail -1 | cut -c1-150)"
  python3 - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(f"{sys.argv[1]}/trace/*/*kernel_stats.csv")[0]
for row in list(csv.DictReader(open(f)))[:7]:
    print(f"   {row['Name'][:64]:64s} calls {row['Calls']:>5s} avg {float(row['AverageNs'])/1e6:8.3f} ms  {row['Percentage']} %")
PY
done
