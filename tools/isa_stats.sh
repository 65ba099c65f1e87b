#!/bin/bash
# usage: tools/isa_stats.sh <file.hip> <mangled-kernel-substring>   -- registers, scratch, static instruction counts
set -e
SRC=/root/repo/eeyore_amd/csrc/$1
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-fast-math -S --cuda-device-only -I/root/repo/eeyore_amd/csrc "$SRC" -o /tmp/isa/out.s 2>/dev/null
for k in $(grep -o "^_Z[A-Za-z0-9_]*$2[A-Za-z0-9_]*:" /tmp/isa/out.s | tr -d ':' | sort -u); do
  awk "/^$k:/,/s_endpgm/" /tmp/isa/out.s > /tmp/isa/$k.s
  echo "$k: valu $(grep -c '^\sv_' /tmp/isa/$k.s) branches $(grep -c 's_cbranch' /tmp/isa/$k.s) mfma $(grep -c v_mfma /tmp/isa/$k.s) lds $(grep -c '^\sds_' /tmp/isa/$k.s)"
  grep -A60 "^$k:" /tmp/isa/out.s >/dev/null
  awk "/^$k:/,/Occupancy/" /tmp/isa/out.s | grep -i "ScratchSize\|; NumVgprs\|Occupancy" | tr '\n' ' '; echo
done
