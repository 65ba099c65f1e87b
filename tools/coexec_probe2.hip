// (a) cycles per v_mfma_f32_4x4x1 with 1/2/4 independent accumulators; (b) does v_exp_f32 (transcendental) from the
// partner wave co-execute with f32 MFMA?  (c) VALU fma issue with 1 vs 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(256, 1) k44(int iters, float* out) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  f32x4 d[4];
  for (int i = 0; i < 4; ++i) d[i] = (f32x4){0, 0, 0, 0};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 128; ++u) d[u % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d[u % NACC], 0, 0, 0);
  }
  out[blockIdx.x * 256 + threadIdx.x] = d[0][0] + d[1][0] + d[2][0] + d[3][0];
}

// waves 0-3: 32x32x2 chain; waves 4-7: v_exp loop (mode as in coexec_probe)
__global__ void __launch_bounds__(512, 2) ktrans(int mode, int iters, float* out) {
  const int wave = threadIdx.x >> 6;
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  if (wave < 4) {
    if (mode == 1) return;
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c[r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  } else {
    if (mode == 0) return;
    float v[8];
    for (int r = 0; r < 8; ++r) v[r] = a + r;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = __builtin_amdgcn_exp2f(v[r]);
      }
    }
    float s = 0;
    for (int r = 0; r < 8; ++r) s += v[r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  }
}

// VALU fma only, W waves per SIMD (blocks of 256*W threads, one block per CU)
template <int W>
__global__ void __launch_bounds__(256 * W, W) kfma(int iters, float* out) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
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
  out[blockIdx.x * 256 * W + threadIdx.x] = s;
}

template <typename F>
float timeit(F f) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 1024 * 4);
  const int iters = 2000;
  float t1 = timeit([&] { k44<1><<<256, 256>>>(iters, out); });
  float t2 = timeit([&] { k44<2><<<256, 256>>>(iters, out); });
  float t4 = timeit([&] { k44<4><<<256, 256>>>(iters, out); });
  printf("4x4x1 x 256000 per wave: 1 acc %.3f ms, 2 acc %.3f ms, 4 acc %.3f ms\n", t1, t2, t4);
  for (int mode = 0; mode < 3; ++mode) {
    float t = timeit([&] { ktrans<<<256, 512>>>(mode, iters, out); });
    printf("32x32x2 (32000/wave) + v_exp (256000/wave) mode %d: %.3f ms\n", mode, t);
  }
  float f1 = timeit([&] { kfma<1><<<256, 256>>>(iters, out); });
  float f2 = timeit([&] { kfma<2><<<256, 512>>>(iters, out); });
  printf("fma x 512000 per wave: 1 wave/SIMD %.3f ms, 2 waves/SIMD %.3f ms\n", f1, f2);
  return 0;
}
