#!/bin/bash
# Where the fused 16x16x4 kernels spend their time: whole-library builds with parts of the tile loop compiled out
# (F16_ABLATE in ey_fused16.hip; results are wrong in such builds, only the timing is read).
#   here:            tools/ablate_fused16.sh build
#   on the GPU box:  tools/ablate_fused16.sh run > gpurun_out/ablate_f16.txt
# columns of tools/ab_fused16.py: 4-32-32-3 f64, 4-16-16-3 f32, 4-32-32-3 tanh f32, 4-64-64-3 f32 (leapfrog-steps/s x chains)
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  tools/build_variants.sh "f16a0=-DF16_ABLATE=0" "f16a1=-DF16_ABLATE=1" "f16a2=-DF16_ABLATE=2" "f16a4=-DF16_ABLATE=4" \
    "f16a8=-DF16_ABLATE=8" "f16a15=-DF16_ABLATE=15"
else
  for v in f16a0 f16a1 f16a2 f16a4 f16a8 f16a15 f16a0; do
    EEYORE_AMD_LIB=tools/abl/lib_$v.so python tools/ab_fused16.py 2>&1 | grep -v amdgpu.ids
  done
fi
