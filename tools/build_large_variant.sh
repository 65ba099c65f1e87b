#!/bin/bash
# tools/build_large_variant.sh NAME "extra compiler flags"  -> tools/abl/lib_NAME.so (ey_large.hip rebuilt with the flags,
# the library's other objects as they are)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); O=$ROOT/eeyore_amd/lib/obj; mkdir -p $ROOT/tools/abl /tmp/lv_$1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -w $2 -c $ROOT/eeyore_amd/csrc/ey_large.hip -o /tmp/lv_$1/ey_large.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/tools/abl/lib_$1.so /tmp/lv_$1/ey_large.o $O/ey_api.o $O/ey_generic.o $O/ey_mfma32.o $O/ey_fused16.o $O/ey_stats.o
echo built tools/abl/lib_$1.so
