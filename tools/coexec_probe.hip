// Can f32 MFMA (v_mfma_f32_32x32x2_f32 / 4x4x1) co-execute with VALU / transcendental work of the partner wave on the
// same SIMD?  512-thread blocks: waves 0-3 run an MFMA chain, waves 4-7 a VALU loop.  mode 0: MFMA only, 1: VALU only,
// 2: both.  If both together take ~max(t0, t1) they overlap; if ~t0 + t1 they do not.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ void __launch_bounds__(512, 2) k(int mode, int iters, float* out) {
  const int wave = threadIdx.x >> 6;
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  if (wave < 4) {
    if (mode == 1) return;
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0;
    f32x4 d = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
      if (KIND == 0) {
#pragma unroll
        for (int u = 0; u < 16; ++u) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
      } else {
#pragma unroll
        for (int u = 0; u < 128; ++u) d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d, 0, 0, 0);
      }
    }
    float s = d[0];
    for (int r = 0; r < 16; ++r) s += c[r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  } else {
    if (mode == 0) return;
    float v[8];
    for (int r = 0; r < 8; ++r) v[r] = a + r;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 32; ++u) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = __builtin_fmaf(v[r], b, 0.5f);
      }
    }
    float s = 0;
    for (int r = 0; r < 8; ++r) s += v[r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  }
}

template <int KIND>
void run(const char* name) {
  float* out;
  hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  for (int mode = 0; mode < 3; ++mode) {
    k<KIND><<<256, 512>>>(mode, iters, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND><<<256, 512>>>(mode, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // MFMA cycles per wave: iters * 16 * 64 (32x32x2) or iters * 128 * 8 (4x4x1); VALU: iters*256 fma * 4 cyc issue
    printf("%s mode %d: %.3f ms\n", name, mode, ms);
  }
}
int main() {
  run<0>("32x32x2 + fma");
  run<1>("4x4x1   + fma");
  return 0;
}
