#!/bin/bash
# Fabric bytes per dispatch (FETCH_SIZE / WRITE_SIZE passes, tools/pmc_by_kernel.py) of config 5's kernels under several
# settings of one environment variable.  usage (GPU box, repo root): tools/fabric_cfg5_env.sh NAME value value ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
name=$1; shift
for v in "$@"; do
  out=gpurun_out/fab_${name}_$v; rm -rf $out; mkdir -p $out
  export $name=$v
  CMD="python3 tools/bench_config5.py ${AB_CHAINS:-4096} 1"
  rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/c -- $CMD > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/d -- $CMD > /dev/null 2>&1
  echo "== $name=$v"
  python3 tools/pmc_by_kernel.py $out | grep "fabric bytes" | head -4
done
