// k_fused16<double, 32, 4, *> -- the fused kernels of the reference's default dtype (eeyore/models/model.py:7) at hidden widths
// 17..32, among them the headline model in f64 -- as a translation unit of their own, so that the Makefile can build them
// with -mllvm -amdgpu-mfma-vgpr-form (see EY_F16_PART in ey_fused16.hip).  The source is that file.
#define EY_F16_PART 1
#include "ey_fused16.hip"
