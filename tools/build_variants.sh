#!/bin/bash
# Build whole-library A/B variants into tools/abl/lib_<tag>.so: each argument is "tag=-DFLAG=... -DOTHER=..." (or a bare
# number N, short for "vN=-DEY_V=N").  Built here (hipcc cross-compiles), run on the GPU box with tools/ab_libs.sh.
cd "$(dirname "$0")/.."
mkdir -p tools/abl
for a in "$@"; do
  if [[ "$a" == *=* ]]; then tag="${a%%=*}"; flags="${a#*=}"; else tag="v$a"; flags="-DEY_V=$a"; fi
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math $flags -shared \
    -o tools/abl/lib_$tag.so eeyore_amd/csrc/*.hip 2> tools/abl/err_$tag.txt || echo "BUILD FAILED: $tag" &
done
wait
grep -l "error:" tools/abl/err_*.txt
ls -la tools/abl/*.so
