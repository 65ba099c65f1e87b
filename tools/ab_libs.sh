#!/bin/bash
# A/B whole library builds (e.g. tools/abl/lib_prev.so against eeyore_amd/lib/libeeyore_amd.so): separate processes,
# interleaved rounds, the fused HMC kernel alone (tools/ab_variants.py, variant 0).
# usage: [AB_CHAINS="4096 3000"] tools/ab_libs.sh libA libB ...
for chains in ${AB_CHAINS:-4096}; do
  for round in 1 2; do
    for lib in "$@"; do
      echo -n "$chains chains, $lib: "
      EEYORE_AMD_LIB=$lib AB_CHAINS=$chains AB_STEP=0.024 python tools/ab_variants.py 0 | grep "variant 0"
    done
  done
done
