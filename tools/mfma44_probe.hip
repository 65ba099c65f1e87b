// v_mfma_f64_4x4x4_4b_f64 on gfx950: (a) cycles per instruction with 1 / 4 independent accumulators at one and two waves
// per SIMD, beside v_mfma_f64_16x16x4_f64; (b) which A lane and which B lane feed each D lane (one register per lane).
//   hipcc -O3 --offload-arch=gfx950 tools/mfma44_probe.hip -o tools/mfma44_probe && tools/mfma44_probe   (on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int NACC, int W, bool SMALL>
__global__ void __launch_bounds__(256 * W, W) k_rate(int iters, double* out) {
  double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
  double s[NACC];
  f64x4 q[NACC];
  for (int k = 0; k < NACC; ++k) { s[k] = 0; q[k] = f64x4{0, 0, 0, 0}; }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int k = 0; k < NACC; ++k) {
        if (SMALL) s[k] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s[k], 0, 0, 0);
        else q[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, q[k], 0, 0, 0);
      }
  }
  double t = 0;
  for (int k = 0; k < NACC; ++k) t += SMALL ? s[k] : q[k][0] + q[k][3];
  out[blockIdx.x * 256 * W + threadIdx.x] = t;
}
__global__ void k_fma(int iters, float* out) {
  float v[8];
  for (int r = 0; r < 8; ++r) v[r] = threadIdx.x * 1e-3f + r;
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = __builtin_fmaf(v[r], 1.0001f, 0.5f);
  float s = 0;
  for (int r = 0; r < 8; ++r) s += v[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ void k_map(const double* a, const double* b, double* d) {
  const int l = threadIdx.x;
  d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}
template <typename F>
static float timeit(F f) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  double* out; float* outf;
  (void)hipMalloc(&out, 256 * 512 * 8); (void)hipMalloc(&outf, 256 * 256 * 4);
  const int iters = 2000;
  const float f1 = timeit([&] { k_fma<<<256, 256>>>(iters, outf); });
  const double hz = 2.22 / (f1 * 1e-3 / (iters * 128.0));
  auto cyc = [&](float ms, double n) { return ms * 1e-3 * hz / n; };
  printf("(a) cycles per MFMA of one wave's stream (two waves per SIMD: per instruction of the SIMD)\n");
#define ROW(NAME, NACC, SMALL) \
  printf("  %-34s 1 wave/SIMD %6.1f   2 waves/SIMD %6.1f\n", NAME, \
         cyc(timeit([&] { k_rate<NACC, 1, SMALL><<<256, 256>>>(iters, out); }), iters * 16.0 * NACC), \
         cyc(timeit([&] { k_rate<NACC, 2, SMALL><<<256, 512>>>(iters, out); }), iters * 16.0 * NACC * 2))
  ROW("4x4x4_4b f64, 1 accumulator", 1, true);
  ROW("4x4x4_4b f64, 4 accumulators", 4, true);
  ROW("16x16x4 f64, 1 accumulator", 1, false);
  ROW("16x16x4 f64, 4 accumulators", 4, false);
  // lane maps: D lane l = sum over the A / B lanes that feed it
  double *a, *b, *d;
  (void)hipMalloc(&a, 512); (void)hipMalloc(&b, 512); (void)hipMalloc(&d, 512);
  std::vector<double> ha(64), hb(64), hd(64);
  printf("(b) D lane <- lists of (A lane, B lane) pairs with a non-zero product\n");
  std::vector<std::vector<std::pair<int, int>>> feed(64);
  for (int la = 0; la < 64; ++la) {
    for (int l = 0; l < 64; ++l) { ha[l] = l == la ? 1 : 0; hb[l] = l + 1; }
    (void)hipMemcpy(a, ha.data(), 512, hipMemcpyHostToDevice); (void)hipMemcpy(b, hb.data(), 512, hipMemcpyHostToDevice);
    k_map<<<1, 64>>>(a, b, d);
    (void)hipMemcpy(hd.data(), d, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l)
      if (hd[l] != 0) feed[l].push_back({la, (int)hd[l] - 1});
  }
  for (int l = 0; l < 64; ++l) {
    printf("  D lane %2d:", l);
    for (auto& p : feed[l]) printf(" (A%2d,B%2d)", p.first, p.second);
    printf("\n");
  }
  return 0;
}
