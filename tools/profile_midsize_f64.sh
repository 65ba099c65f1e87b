#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/midsize64
sed 's/torch.float32/torch.float64/g; s/np.float32/np.float64/g' /tmp/midsize.py > /tmp/midsize64.py
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 /tmp/midsize64.py > $OUT.log 2>&1
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
head -10 $OUT/kernel_stats.csv | cut -c1-110
