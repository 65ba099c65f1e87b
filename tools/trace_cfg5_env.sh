#!/bin/bash
# Per-kernel times (rocprofv3 kernel trace) of config 5's share under several settings of ONE environment variable.
# usage (GPU box, repo root): tools/trace_cfg5_env.sh NAME value value ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
name=$1; shift
for v in "$@"; do
  out=gpurun_out/tre_${name}_$v; rm -rf $out; mkdir -p $out
  export $name=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/bench_config5.py ${AB_CHAINS:-4096} ${AB_ITERS:-1} > $out/run.log 2>&1
  echo "== $name=$v: $(grep -h 'leapfrog' $out/run.log | tail -1 | cut -c1-150)"
  python3 - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(f"{sys.argv[1]}/trace/*/*kernel_stats.csv")[0]
for row in list(csv.DictReader(open(f)))[:7]:
    print(f"   {row['Name'][:64]:64s} calls {row['Calls']:>5s} avg {float(row['AverageNs'])/1e6:8.3f} ms  {row['Percentage']} %")
PY
done
