// How fast a CU retires the stores of a 128 x 128 f32 output tile: as the 32x32 MFMA accumulator layout gives them (64
// global_store_dword per lane, an instruction = 2 rows x 32 consecutive columns) against 16 global_store_dwordx4 per lane (an
// instruction = 2 rows x 128 columns, what a transposition through LDS would allow).  Twelve waves per CU, every workgroup its
// own tiles; row stride = 128 or 100 floats.   hipcc --offload-arch=gfx950 -O3 -o tools/store_probe tools/store_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int WIDE>
__global__ __launch_bounds__(256, 3) void k(float* out, int tiles, int ld, unsigned long long* ticks) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  float v = tid * 0.001f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int t = 0; t < tiles; ++t) {
    float* C = out + ((size_t)blockIdx.x * tiles + t) * 128 * ld;
    if constexpr (!WIDE) {  // the accumulator layout: wave quadrant (wm, wn) 64 x 64, tile (i, j), register r: row 8 (r / 4) + 4 h + r % 4, column c
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = wm * 64 + 32 * i + 8 * (r >> 2) + 4 * h + (r & 3), n = wn * 64 + 32 * j + c;
            if (n < ld) C[m * ld + n] = v + r;
          }
    } else {  // rows of the tile: an instruction covers rows (2 s + h) of this wave's 32, 128 columns as 32 lanes x 16 bytes
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int m = wave * 32 + 2 * s + h, n = 4 * c;
        if (n < ld) *reinterpret_cast<f4u*>(C + m * ld + n) = f4{v, v + 1, v + 2, v + 3};
      }
    }
    v += 1.0f;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0 && (blockIdx.x & 15) == 0) atomicAdd(ticks, t1 - t0);
}

int main() {
  int cus = 0;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  float* out;
  unsigned long long* ticks;
  {
  }
  CK(hipMalloc(&out, (size_t)cus * 3 * 64 * 128 * 128 * 4));
  CK(hipMalloc(&ticks, 8));
  const int tiles = 64;
  for (int frac : {1, 8}) {  // every CU with three workgroups; an eighth of that (the write path far from saturated)
  const int grid = cus * 3 / frac;
  printf("%d workgroups:\n", grid);
  for (int ld : {128, 100}) {
    for (int wide = 0; wide < 2; ++wide) {
      CK(hipMemset(ticks, 0, 8));
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(ticks, 0, 8));
        CK(hipEventRecord(e0));
        if (wide) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, tiles, ld, ticks);
        else hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, tiles, ld, ticks);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
      }
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long t;
      CK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost));
      const double bytes = (double)grid * tiles * 128 * ld * 4;
      printf("row stride %3d floats, %-22s %7.3f ms  %6.2f TB/s  %7.0f cycles per tile and workgroup (three workgroups per CU)\n", ld,
             wide ? "16 x dwordx4 per lane:" : "64 x dword per lane:", ms, bytes / (ms * 1e-3) / 1e12, (double)t / ((grid + 15) / 16) / tiles);
    }
  }
  }
  return 0;
}
