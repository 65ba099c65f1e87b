#!/bin/bash
# A/B whole-library builds on config 5's per-GPU share (tools/bench_config5.py), interleaved in separate processes.
# usage: [AB_ROUNDS=2] tools/ab_cfg5.sh libA libB ...     ("default" = the in-tree library)
for round in $(seq 1 ${AB_ROUNDS:-2}); do
  for lib in "$@"; do
    echo -n "$lib: "
    if [ "$lib" = default ]; then python tools/bench_config5.py ${AB_CHAINS:-4096} 1 2>&1 | tail -1 | cut -c1-170
    else EEYORE_AMD_LIB=$lib python tools/bench_config5.py ${AB_CHAINS:-4096} 1 2>&1 | tail -1 | cut -c1-170; fi
  done
done
