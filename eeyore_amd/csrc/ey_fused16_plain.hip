// k_fused16<T, H, WAVES, V, F16_HMC_PLAIN> for the shapes at two and four waves per SIMD (f32 H = 16 / 32, f64 H = 16): the HMC
// draw with its run-time options compiled out (every parameter under the same prior, no temperature, no tuner attached), which
// is what `HMC.run` on a plain posterior issues.  +4-5 % on these shapes; a translation unit of its own so that the twelve
// kernels build beside the main unit (EY_F16_PART in ey_fused16.hip).  The source is that file.
#define EY_F16_PART 2
#include "ey_fused16.hip"
