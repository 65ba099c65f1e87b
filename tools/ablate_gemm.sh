#!/bin/bash
# Timing-only ablations of the batched GEMM (eeyore_amd/csrc/ey_large.hip, -DBG_ABL bit mask: 1 = no LDS fragment
# reads, 2 = no global fetch / LDS staging).  Built here into tools/abl/, run on the GPU box:
#   for a in 0 1 2 3; do EEYORE_AMD_LIB=tools/abl/lib_bg$a.so python3 tools/bench_config5.py 1024 3; done
# Results of such builds are wrong; only their timing is read.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/abl
for a in 0 1 2 3; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -DBG_ABL=$a -shared \
    -o tools/abl/lib_bg$a.so eeyore_amd/csrc/*.hip 2> /dev/null &
done
wait
ls -la tools/abl/lib_bg*
