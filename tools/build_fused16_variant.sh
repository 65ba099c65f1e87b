#!/bin/bash
# tools/build_fused16_variant.sh NAME "extra compiler flags" -> tools/abl/lib_NAME.so (ey_fused16.hip rebuilt with the flags through the
# Makefile's pipeline incl. the MFMA load-hazard pass, the library's other objects as they are)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); O=$ROOT/eeyore_amd/lib/obj; T=/tmp/fv_$1; mkdir -p $ROOT/tools/abl $T
LLVM=/opt/rocm/lib/llvm/bin
CXX="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -w"
/opt/rocm/bin/hipcc $CXX $2 --offload-device-only -S ${SRC:-$ROOT/eeyore_amd/csrc/ey_fused16.hip} -o $T/dev.s
python3 $ROOT/eeyore_amd/csrc/mfma_load_hazard.py $T/dev.s $T/fixed.s > /dev/null
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $T/fixed.s -o $T/dev.o
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $T/f.co $T/dev.o
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$T/f.co -output=$T/f.hipfb
/opt/rocm/bin/hipcc $CXX $2 --offload-host-only -Xclang -fcuda-include-gpubinary -Xclang $T/f.hipfb -c ${SRC:-$ROOT/eeyore_amd/csrc/ey_fused16.hip} -o $T/ey_fused16.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/tools/abl/lib_$1.so $T/ey_fused16.o $O/ey_fused16_d32.o $O/ey_fused16_plain.o $O/ey_api.o $O/ey_generic.o $O/ey_mfma32.o $O/ey_large.o $O/ey_stats.o
echo built tools/abl/lib_$1.so
