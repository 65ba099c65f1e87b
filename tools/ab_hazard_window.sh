#!/bin/bash
# Builds of the library that differ only in what eeyore_amd/csrc/mfma_load_hazard.py does to ey_fused16's assembly
# (none / window 8 / window 18 wait states behind v_mfma_f64_16x16x4), for tools/ab_fused16.py on a GPU box:
#   tools/ab_hazard_window.sh build     (here)        tools/ab_hazard_window.sh run   (on the box)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
O=$ROOT/eeyore_amd/lib/obj; A=$ROOT/tools/abl; LL=/opt/rocm/lib/llvm/bin; SRC=$ROOT/eeyore_amd/csrc/ey_fused16.hip
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -w"
if [ "$1" = build ]; then
  mkdir -p $A
  for w in 0 8 18; do
    d=$A/hz$w; mkdir -p $d
    if [ $w = 0 ]; then cp $O/ey_fused16.dev.s $d/f.s; else EY_HAZARD_W16=$w python3 $ROOT/eeyore_amd/csrc/mfma_load_hazard.py $O/ey_fused16.dev.s $d/f.s; fi
    $LL/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $d/f.s -o $d/dev.o
    $LL/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $d/f.co $d/dev.o
    $LL/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$d/f.co -output=$d/f.hipfb
    /opt/rocm/bin/hipcc $FL --offload-host-only -Xclang -fcuda-include-gpubinary -Xclang $d/f.hipfb -c $SRC -o $d/f16.o
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $A/lib_hz$w.so $d/f16.o $O/ey_api.o $O/ey_generic.o $O/ey_mfma32.o $O/ey_large.o $O/ey_stats.o
    rm -rf $d
  done
  ls -la $A/lib_hz*.so
else
  for round in 1 2 3; do for w in 0 8 18; do EEYORE_AMD_LIB=$A/lib_hz$w.so python $ROOT/tools/ab_fused16.py; done; done
fi
