#!/bin/bash
# A/B one library under different values of an environment variable (a launch-time knob read by the host code):
# separate processes, interleaved rounds, the fused HMC kernel alone (tools/ab_variants.py, variant 0).
# usage: [AB_CHAINS=4096] [AB_ROUNDS=2] [AB_IPL=25] tools/ab_env.sh VAR value1 value2 ...
var=$1; shift
for round in $(seq 1 ${AB_ROUNDS:-2}); do
  for val in "$@"; do
    echo -n "$var=$val: "
    env $var=$val AB_CHAINS=${AB_CHAINS:-4096} AB_STEP=0.024 AB_IPL=${AB_IPL:-25} python tools/ab_variants.py 0 | grep "variant 0\|checksum" | tr '\n' ' '
    echo
  done
done
