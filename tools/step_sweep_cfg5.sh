#!/bin/bash
# acceptance of config 5's HMC (L = 20, MLP(784-128-10), N = 1024) against the step, after a burn-in at that step
for s in ${STEPS:-0.001 0.002 0.003 0.004 0.006 0.008}; do
  EY_CFG5_STEP=$s EY_CFG5_BURNIN=${BURNIN:-40} python tools/bench_config5.py ${CHAINS:-256} 10 2>&1 | tail -1 | sed 's/.*MFMA peak), //'
done
