#!/bin/bash
# Kernel trace of the layerwise path on a mid-size model that fits LDS (MLP(10-100-10), N = 256, 2048 chains, HMC L = 10;
# six iterations).  usage: [DIMS=20,100,100,5 N=512 C=1024] tools/profile_midsize.sh [f32|f64]   (on the GPU box, from the repo root)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
DT=${1:-f32}
DIMS=${DIMS:-10,100,10}
N=${N:-256}
C=${C:-2048}
OUT=gpurun_out/midsize_${DT}_${DIMS//,/-}
mkdir -p gpurun_out
cat > /tmp/midsize_$DT.py <<PY
import sys, numpy as np, torch
sys.path.insert(0, '.')
from eeyore_amd.plan import Plan
dev = torch.device('cuda', 0)
tdt, ndt = (torch.float64, np.float64) if "$DT" == "f64" else (torch.float32, np.float32)
dims, N, C = [$DIMS], $N, $C
rng = np.random.default_rng(0)
x = rng.standard_normal((N, dims[0])).astype(ndt)
y = np.eye(dims[-1], dtype=ndt)[rng.integers(0, dims[-1], N)]
K = len(dims) - 1
pl = Plan(dims, [1] * K, [1] * (K - 1) + [0], 1, tdt, dev)
pl.set_data(torch.tensor(x, device=dev), torch.tensor(y, device=dev))
pl.set_prior(torch.zeros(pl.P), torch.ones(pl.P))
pl.set_variant(int('${VARIANT:-0}'))
th = 0.1 * pl.philox_normal(C, seed=0, it=0)
t, g = pl.log_target_grad(th)
for i in range(6):
    pl.hmc_step(th, t, g, 0.005, 10, seed=1, it=1 + i)
torch.cuda.synchronize()
print(pl.kernel)
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 /tmp/midsize_$DT.py > $OUT.log 2>&1
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
head -14 $OUT/kernel_stats.csv | cut -c1-150
