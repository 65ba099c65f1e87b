#!/bin/bash
# The GPU suite on the diagnostic builds of `make spill` (one translation unit compiled with -vgpr-regalloc=fast):
# a unit whose tests pass here does not depend on where the register allocator puts its spills.
# usage (on the GPU box): bash tools/spill_suites.sh [unit ...]     logs: gpurun_out/spill_<unit>.log
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
units=${@:-ey_fused16 ey_fused16_d32 ey_fused16_plain ey_mfma32 ey_generic ey_large}
for u in $units; do
  lib=eeyore_amd/lib/libeeyore_amd_spill_$u.so
  [ -f "$lib" ] || { echo "$lib missing (make -C eeyore_amd/csrc spill)"; exit 2; }
  echo "== $u"
  EEYORE_AMD_LIB=$lib timeout -k 10 ${SPILL_TIMEOUT:-420} python -m pytest tests -m gpu -q -p no:cacheprovider ${SPILL_PYTEST:-} > gpurun_out/spill_$u.log 2>&1
  rc=$?
  tail -n 3 gpurun_out/spill_$u.log
  grep -E "^(FAILED|ERROR)" gpurun_out/spill_$u.log | cut -c1-200
  # a step that was killed at its limit ends the call: no further GPU step behind it
  if [ $rc -ge 124 ]; then echo "$u: killed at its limit (rc $rc), stopping"; exit $rc; fi
done
