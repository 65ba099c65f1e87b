// The pipelined form of the fused f32 trajectory kernel (eval_pipe / k_mfma32p in ey_mfma32.hip) as a translation unit of its
// own: one wave per SIMD, and built without packed f32 instructions (see the Makefile).
#define EY_MF_PART 1
#include "ey_mfma32.hip"
