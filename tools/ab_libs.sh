#!/bin/bash
# A/B whole library builds (e.g. tools/abl/lib_prev.so against eeyore_amd/lib/libeeyore_amd.so): separate processes,
# interleaved rounds, the fused HMC kernel alone (tools/ab_variants.py, variant 0).  The checksum printed with each
# run (sum of theta and of the log-targets after the same sequence of draws) must agree between builds whose
# arithmetic is meant to be identical.
# usage: [AB_CHAINS="4096 3000"] [AB_ROUNDS=2] tools/ab_libs.sh libA libB ...
for chains in ${AB_CHAINS:-4096}; do
  for round in $(seq 1 ${AB_ROUNDS:-2}); do
    for lib in "$@"; do
      echo -n "$chains chains, $lib: "
      EEYORE_AMD_LIB=$lib AB_CHAINS=$chains AB_STEP=0.024 AB_IPL=${AB_IPL:-1} python tools/ab_variants.py 0 | grep "variant 0\|checksum" | tr '\n' ' '
      echo
    done
  done
done
