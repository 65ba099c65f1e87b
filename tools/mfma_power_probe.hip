// What the matrix pipe sustains at the socket's power limit: a loop of nothing but v_mfma_f32_32x32x16_bf16 (or
// v_mfma_f32_32x32x2_f32) with four independent accumulators per wave and operands whose bits change every iteration
// (a pipe fed zeros draws less), 2 or 3 waves per SIMD, several seconds -- run beside tools/clock_watch.sh for the clock.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_power_probe tools/mfma_power_probe.hip && tools/mfma_power_probe [seconds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool BF16>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned seed) {
  f16v acc[4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  unsigned x = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u) ^ seed;
  for (int i = 0; i < iters; ++i) {
    x = x * 1664525u + 1013904223u;  // one vector instruction per four products
    if constexpr (BF16) {
      // bf16 values in [1, 2): exponent 0x3f8, mantissa bits from x
      const unsigned w0 = 0x3f803f80u | (x & 0x007f007fu), w1 = 0x3f803f80u | ((x >> 7) & 0x007f007fu);
      const u4 ua = {w0, w1, w0 ^ 0x00110022u, w1 ^ 0x00440008u}, ub = {w1, w0, w1 ^ 0x00210003u, w0 ^ 0x00050041u};
      const bf8 a = __builtin_bit_cast(bf8, ua), b = __builtin_bit_cast(bf8, ub);
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
    } else {
      const float a = __builtin_bit_cast(float, 0x3f800000u | (x & 0x007fffffu)), b = __builtin_bit_cast(float, 0x3f800000u | ((x >> 3) & 0x007fffffu));
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k], 0, 0, 0);
    }
    if ((i & 1023) == 1023) {  // keep the sums finite
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] *= 1e-6f;
    }
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[k][r];
  if (s == 12345.678f) out[0] = s;
}

template <bool BF16>
static void run(const char* name, int waves_per_simd, double seconds, double flop_per_mfma) {
  int cus = 0;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  float* out;
  CK(hipMalloc(&out, 4));
  const int grid = cus * waves_per_simd, iters = 200000;  // 256 threads = 4 waves = one per SIMD; waves_per_simd workgroups per CU
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<BF16>, dim3(grid), dim3(256), 0, 0, out, 1000, 1u);
  CK(hipDeviceSynchronize());
  double total_ms = 0, last = 0;
  int launches = 0;
  while (total_ms < seconds * 1e3) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<BF16>, dim3(grid), dim3(256), 0, 0, out, iters, (unsigned)launches);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    total_ms += ms; last = ms; ++launches;
  }
  const double mfmas = (double)grid * 4 /*waves*/ * iters * 4.0;
  printf("%-28s %d wave(s) per SIMD: %7.1f TFLOP/s over the last launch (%.0f ms), %d launches in %.1f s\n", name, waves_per_simd,
         mfmas * flop_per_mfma / (last * 1e-3) / 1e12, last, launches, total_ms * 1e-3);
  CK(hipFree(out));
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 4.0;
  run<true>("v_mfma_f32_32x32x16_bf16", 2, seconds, 2.0 * 32 * 32 * 16);
  run<true>("v_mfma_f32_32x32x16_bf16", 3, seconds, 2.0 * 32 * 32 * 16);
  run<false>("v_mfma_f32_32x32x2_f32", 2, seconds, 2.0 * 32 * 32 * 2);
  return 0;
}
