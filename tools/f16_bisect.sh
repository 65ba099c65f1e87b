#!/bin/bash
# Diagnostic builds of ey_fused16.hip with ONE kernel instantiation (seconds to compile), linked with the library's other
# objects, each run through tools/f16_check.py on a GPU box.  One line per variant in gpurun_out/f16_bisect.txt.
#   tools/f16_bisect.sh SIZE H V "dims" "acts" lik tag N  variants-file
# variants-file: one variant per line, "name | compiler flags"
set -u
SIZE=$1; H=$2; V=$3; DIMS=$4; ACTS=$5; LIK=$6; TAG=$7; N=$8; VARIANTS=$9
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=$ROOT/eeyore_amd/lib/obj
WORK=${WORK:-/tmp/f16_bisect}
OUT=$ROOT/gpurun_out/${OUTNAME:-f16_bisect.txt}
mkdir -p "$WORK" "$ROOT/gpurun_out"
BASE="-std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -w -DF16_ONLY_SIZE=$SIZE -DF16_ONLY_H=$H -DF16_ONLY_V=$V"
echo "# instantiation size=$SIZE H=$H V=$V case $DIMS $ACTS lik=$LIK $TAG N=$N" >> "$OUT"
while IFS='|' read -r name flags; do
  name=$(echo "$name" | xargs); [ -z "$name" ] && continue
  case "$name" in \#*) continue;; esac
  d=$WORK/$name; mkdir -p "$d"
  if ! ( cd "$d" && /opt/rocm/bin/hipcc $BASE $flags -c "$ROOT/eeyore_amd/csrc/ey_fused16.hip" -o f16.o > build.log 2>&1 ); then
    echo "$name | BUILD-FAILED | $(tail -1 "$d/build.log")" >> "$OUT"; continue
  fi
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$d/lib.so" "$d/f16.o" "$OBJ/ey_api.o" "$OBJ/ey_generic.o" \
    "$OBJ/ey_mfma32.o" "$OBJ/ey_large.o" "$OBJ/ey_stats.o" >> "$d/build.log" 2>&1 || { echo "$name | LINK-FAILED" >> "$OUT"; continue; }
  EEYORE_AMD_LIB=$d/lib.so timeout -k 10 120 python "$ROOT/tools/f16_check.py" "$DIMS" "$ACTS" "$LIK" "$TAG" "$N" > "$d/check.log" 2>&1
  rc=$?
  echo "$name | rc=$rc | $(tail -1 "$d/check.log") | flags: $flags" >> "$OUT"
  grep WRONG "$d/check.log" | sed "s/^/      /" >> "$OUT"
  [ $rc -ge 124 ] && { echo "timeout: stopping" >> "$OUT"; exit 1; }
done < "$VARIANTS"
exit 0
