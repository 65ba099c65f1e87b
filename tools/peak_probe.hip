// Measured ceilings of one MI355X, to quote beside the datasheet peaks: the f32 matrix rate of
// v_mfma_f32_32x32x2_f32 with independent accumulators, and HBM stream read / copy bandwidth.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/peak_probe tools/peak_probe.hip && /tmp/peak_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma_peak(float* out, int iters) {
  f16v acc[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NACC; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[k][r];
  if (s == 12345.678f) out[0] = s;  // never true: keeps the accumulators live
}

// 16x16x4 tiles (the fused kernel for other widths and for f64 is built on them): T = float or double
typedef double d4v __attribute__((ext_vector_type(4)));
template <typename T, int NACC>
__global__ __launch_bounds__(256) void k_mfma16_peak(T* out, int iters) {
  typedef T v4 __attribute__((ext_vector_type(4)));
  v4 acc[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a) acc[a] = (v4){0, 0, 0, 0};
  T a = threadIdx.x * (T)1e-3, b = blockIdx.x * (T)1e-3 + (T)1;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < NACC; ++k) {
      if constexpr (sizeof(T) == 8) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k], 0, 0, 0);
      else acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k], 0, 0, 0);
    }
  }
  T s = 0;
#pragma unroll
  for (int k = 0; k < NACC; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  if (s == (T)12345.678) out[0] = s;
}

__global__ __launch_bounds__(256) void k_read(const f4v* __restrict__ x, size_t n, float* out) {
  f4v s = {0, 0, 0, 0};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    s += __builtin_nontemporal_load(x + i);
  }
  if (s.x + s.y + s.z + s.w == 12345.678f) out[0] = s.x;
}

__global__ __launch_bounds__(256) void k_copy(const f4v* __restrict__ x, f4v* __restrict__ y, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(__builtin_nontemporal_load(x + i), y + i);
}

template <class F>
static float time_ms(F&& launch, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("%s: %d CUs, %d MHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000);
  float* out;
  CK(hipMalloc(&out, 64));
  const int iters = 4000;
  auto report = [&](const char* name, int nacc, float ms, int blocks) {
    double flops = 2.0 * 32 * 32 * 2 * (double)nacc * iters * 4 /*waves*/ * blocks;
    printf("%-28s %8.3f ms  %7.1f TFLOP/s\n", name, ms, flops / ms * 1e-9);
  };
  int blocks = prop.multiProcessorCount * 2;  // 8 waves per CU = 2 per SIMD
  report("mfma 32x32x2 f32, 1 acc", 1, time_ms([&] { k_mfma_peak<1><<<blocks, 256>>>(out, iters); }, 5), blocks);
  report("mfma 32x32x2 f32, 2 acc", 2, time_ms([&] { k_mfma_peak<2><<<blocks, 256>>>(out, iters); }, 5), blocks);
  report("mfma 32x32x2 f32, 4 acc", 4, time_ms([&] { k_mfma_peak<4><<<blocks, 256>>>(out, iters); }, 5), blocks);

  auto report16 = [&](const char* name, int nacc, float ms, int nblocks) {
    double flops = 2.0 * 16 * 16 * 4 * (double)nacc * iters * 4 /*waves*/ * nblocks;
    printf("%-28s %8.3f ms  %7.1f TFLOP/s\n", name, ms, flops / ms * 1e-9);
  };
  double* outd;
  CK(hipMalloc(&outd, 64));
  report16("mfma 16x16x4 f32, 1 acc", 1, time_ms([&] { k_mfma16_peak<float, 1><<<blocks, 256>>>(out, iters); }, 5), blocks);
  report16("mfma 16x16x4 f32, 4 acc", 4, time_ms([&] { k_mfma16_peak<float, 4><<<blocks, 256>>>(out, iters); }, 5), blocks);
  report16("mfma 16x16x4 f64, 1 acc", 1, time_ms([&] { k_mfma16_peak<double, 1><<<blocks, 256>>>(outd, iters); }, 5), blocks);
  report16("mfma 16x16x4 f64, 4 acc", 4, time_ms([&] { k_mfma16_peak<double, 4><<<blocks, 256>>>(outd, iters); }, 5), blocks);
  report16("mfma 16x16x4 f64, 4 acc, 1 wave/SIMD", 4,
           time_ms([&] { k_mfma16_peak<double, 4><<<blocks / 2, 256>>>(outd, iters); }, 5), blocks / 2);

  size_t bytes = (size_t)4 << 30, n = bytes / 16;
  f4v *x, *y;
  CK(hipMalloc(&x, bytes)); CK(hipMalloc(&y, bytes));
  CK(hipMemset(x, 1, bytes)); CK(hipMemset(y, 0, bytes));
  int g = prop.multiProcessorCount * 16;
  float ms = time_ms([&] { k_read<<<g, 256>>>(x, n, out); }, 10);
  printf("%-28s %8.3f ms  %7.1f GB/s\n", "HBM read 4 GiB", ms, bytes / ms * 1e-6);
  ms = time_ms([&] { k_copy<<<g, 256>>>(x, y, n); }, 10);
  printf("%-28s %8.3f ms  %7.1f GB/s (read+write)\n", "HBM copy 4 GiB", ms, 2.0 * bytes / ms * 1e-6);
  return 0;
}
