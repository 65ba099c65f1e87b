#!/bin/bash
# Build diagnostic variants of the library with parts of the MFMA kernel removed (EY_ABLATE bit mask) into
# tools/abl/ (built here, run on the GPU box with tools/ablate_run.sh).  Timing only: results of such builds are wrong.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/abl
for a in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -DEY_ABLATE=$a -shared \
    -o tools/abl/lib_$a.so eeyore_amd/csrc/*.hip 2> /dev/null &
done
wait
ls -la tools/abl
