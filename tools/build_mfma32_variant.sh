#!/bin/bash
# Build A/B variants of ey_mfma32.hip alone, linked with the library's other objects (eeyore_amd/lib/obj/*.o, built by make):
# each argument is "tag=-DFLAG=... [@NOPK]"; result tools/abl/lib_<tag>.so.  About a minute per variant, built in parallel.
# The device side goes through the Makefile's pipeline (assembly kept as tools/abl/mf_<tag>.s); the token @NOPK compiles the
# device code without packed f32 instructions (-target-feature -packed-fp32-ops).
cd "$(dirname "$0")/.."
mkdir -p tools/abl
LLVM=/opt/rocm/lib/llvm/bin
OTHERS=$(ls eeyore_amd/lib/obj/ey_*.o | grep -v "ey_mfma32.o\|dev.o")
CXX="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -w"
for a in "$@"; do
  tag="${a%%=*}"; flags="${a#*=}"; [[ "$a" == *=* ]] || flags=""
  dev=""; [[ "$flags" == *@NOPK* ]] && dev="-Xclang -target-feature -Xclang -packed-fp32-ops"
  flags="${flags//@NOPK/}"
  T=tools/abl/mf_$tag
  ( /opt/rocm/bin/hipcc $CXX $flags $dev --offload-device-only -S eeyore_amd/csrc/ey_mfma32.hip -o $T.s 2> tools/abl/err_$tag.txt && \
    $LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $T.s -o $T.dev.o && \
    $LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $T.co $T.dev.o && \
    $LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 \
      -input=/dev/null -input=$T.co -output=$T.hipfb && \
    /opt/rocm/bin/hipcc $CXX $flags --offload-host-only -Xclang -fcuda-include-gpubinary -Xclang $T.hipfb -c eeyore_amd/csrc/ey_mfma32.hip -o $T.o 2>> tools/abl/err_$tag.txt && \
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/abl/lib_$tag.so $T.o $OTHERS ) || echo "BUILD FAILED: $tag" &
done
wait
ls -la tools/abl/*.so | awk '{print $5, $6, $7, $8, $9}'
