#!/bin/bash
# The measurement set committed under profiles/ for a round (run on the GPU box from the repo root):
#   kernel trace of the default bench, the bench JSON lines (profiled, unprofiled, with the driver's arguments),
#   the PMC passes of the dominant kernel, the issue-cost probe.   usage: tools/evidence.sh <tag>
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r2}
OUT=gpurun_out/evidence_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-compare > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/bench_kernel_stats.csv
echo "trace done"
python3 bench.py > $OUT/bench_unprofiled.json 2> $OUT/bench_unprofiled.err
echo "default bench done"
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err
echo "driver-args bench done"
bash tools/pmc_passes.sh $TAG > $OUT/pmc_passes.log 2>&1
cp gpurun_out/pmc_$TAG/summary.txt $OUT/pmc_summary.txt
cp gpurun_out/pmc_$TAG/pmc_latest.json $OUT/pmc_latest.json
echo "pmc done"
hipcc -O3 --offload-arch=gfx950 -o /tmp/issue_probe tools/issue_probe.hip 2>/dev/null
/tmp/issue_probe > $OUT/issue_probe.txt 2>&1
echo "probe done"
# with the PMC file of THIS tree in place, the bench line carries roofline.traffic (bench.py checks the source hash)
cp $OUT/pmc_latest.json profiles/pmc_latest.json
python3 bench.py > $OUT/bench_with_traffic.json 2>/dev/null || true
tail -c 600 $OUT/bench_unprofiled.json
