// Does vector-ALU work run in the shadow of v_mfma_f32_32x32x16_bf16 when the MFMAs are INDEPENDENT of each other?
// (tools/bf16_coexec_probe.hip measured dependent chains only: there everything adds.)   run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/coexec2_probe.hip -o /tmp/coexec2_probe && /tmp/coexec2_probe
// (d) one stream per wave: [MFMA into accumulator (u mod NACC) + K independent fillers] x 16 per iteration, W waves per SIMD
// (e) two waves per SIMD: an MFMA-only wave with NACC accumulators beside a VALU-only wave
// Times are wall times per iteration in ns at whatever clock the box holds; every line carries its own references
// (the MFMA stream alone, the filler stream alone), so read "both / (mfma + valu)" and "both / max".
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

enum { OP_FMA = 0, OP_EXP = 1, OP_PK = 2, OP_MFMA4 = 3, OP_CVT = 4 };

__device__ __forceinline__ f32x16 mm(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int OP, int K>
__device__ __forceinline__ void fillers(float (&v)[16], f32x4 (&q)[4]) {
#pragma unroll
  for (int r = 0; r < K; ++r) {
    if (OP == OP_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[r & 15]) : "v"(1.0001f));
    if (OP == OP_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v[r & 15]));
    if (OP == OP_PK) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(*reinterpret_cast<double*>(&v[2 * (r & 7)])));
    if (OP == OP_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(v[r & 15]));
    if (OP == OP_MFMA4) q[r & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(v[r & 15], v[(r + 1) & 15], q[r & 3], 0, 0, 0);
  }
}

// MODE 0: MFMAs only, 1: fillers only, 2: both interleaved
template <int NACC, int K, int W, int OP, int MODE>
__global__ void __launch_bounds__(256 * W, W) k_mix(int iters, float* out) {
  u32x4 a = {threadIdx.x * 3u + 0x3f803f80u, 0x3f803f80u, 0x3f003f00u, 0x3e803e80u}, b = {0x3f803f80u, 0x3f003f00u, 0x3f803f80u, 0x3f003f00u};
  f32x16 c[NACC];
  for (int n = 0; n < NACC; ++n)
    for (int r = 0; r < 16; ++r) c[n][r] = 0;
  float v[16];
  f32x4 q[4];
  for (int r = 0; r < 16; ++r) v[r] = threadIdx.x * 1e-3f + r;
  for (int r = 0; r < 4; ++r) q[r] = f32x4{0, 0, 0, 0};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (MODE != 1) c[u % NACC] = mm(a, b, c[u % NACC]);
      if (MODE != 0) fillers<OP, K>(v, q);
    }
  }
  float s = 0;
  for (int n = 0; n < NACC; ++n)
    for (int r = 0; r < 16; ++r) s += c[n][r];
  for (int r = 0; r < 16; ++r) s += v[r];
  for (int r = 0; r < 4; ++r) s += q[r][0] + q[r][1] + q[r][2] + q[r][3];
  out[blockIdx.x * 256 * W + threadIdx.x] = s;
}

// waves 0-3: NACC-accumulator MFMA stream; waves 4-7: filler stream (K x 16 per iteration).  mode 0 / 1 / 2 as above
template <int NACC, int K, int OP>
__global__ void __launch_bounds__(512, 2) k_pair(int mode, int iters, float* out) {
  const int wave = threadIdx.x >> 6;
  if (wave < 4) {
    if (mode == 1) return;
    u32x4 a = {threadIdx.x * 3u + 0x3f803f80u, 0x3f803f80u, 0x3f003f00u, 0x3e803e80u}, b = {0x3f803f80u, 0x3f003f00u, 0x3f803f80u, 0x3f003f00u};
    f32x16 c[NACC];
    for (int n = 0; n < NACC; ++n)
      for (int r = 0; r < 16; ++r) c[n][r] = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) c[u % NACC] = mm(a, b, c[u % NACC]);
    }
    float s = 0;
    for (int n = 0; n < NACC; ++n)
      for (int r = 0; r < 16; ++r) s += c[n][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  } else {
    if (mode == 0) return;
    float v[16];
    f32x4 q[4];
    for (int r = 0; r < 16; ++r) v[r] = threadIdx.x * 1e-3f + r;
    for (int r = 0; r < 4; ++r) q[r] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) fillers<OP, K>(v, q);
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += v[r];
    for (int r = 0; r < 4; ++r) s += q[r][0] + q[r][1] + q[r][2] + q[r][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  }
}

// (f) the same pair with NOPS cycles of s_nop behind every MFMA of the MFMA wave (the wave then does not ask for the vector
// issue port while its MFMA runs), and with s_setprio PM / PV on the MFMA / filler waves
template <int NACC, int K, int OP, int NOPS, int PM, int PV>
__global__ void __launch_bounds__(512, 2) k_pair2(int mode, int iters, float* out) {
  const int wave = threadIdx.x >> 6;
  if (wave < 4) {
    if (mode == 1) return;
    __builtin_amdgcn_s_setprio(PM);
    u32x4 a = {threadIdx.x * 3u + 0x3f803f80u, 0x3f803f80u, 0x3f003f00u, 0x3e803e80u}, b = {0x3f803f80u, 0x3f003f00u, 0x3f803f80u, 0x3f003f00u};
    f32x16 c[NACC];
    for (int n = 0; n < NACC; ++n)
      for (int r = 0; r < 16; ++r) c[n][r] = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        c[u % NACC] = mm(a, b, c[u % NACC]);
        if (NOPS >= 16) asm volatile("s_nop 15");
        if (NOPS >= 32) asm volatile("s_nop 15");
        if (NOPS % 16 == 8) asm volatile("s_nop 7");
        if (NOPS % 16 == 12) asm volatile("s_nop 11");
      }
    }
    float s = 0;
    for (int n = 0; n < NACC; ++n)
      for (int r = 0; r < 16; ++r) s += c[n][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  } else {
    if (mode == 0) return;
    __builtin_amdgcn_s_setprio(PV);
    float v[16];
    f32x4 q[4];
    for (int r = 0; r < 16; ++r) v[r] = threadIdx.x * 1e-3f + r;
    for (int r = 0; r < 4; ++r) q[r] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) fillers<OP, K>(v, q);
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += v[r];
    for (int r = 0; r < 4; ++r) s += q[r][0] + q[r][1] + q[r][2] + q[r][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  }
}

template <typename F>
static float timeit(F f) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  f(); f(); hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  return best;
}
static const char* opname(int op) {
  return op == OP_FMA ? "v_fma_f32" : op == OP_EXP ? "v_exp_f32" : op == OP_PK ? "v_pk_mul_f32" : op == OP_CVT ? "v_cvt_pk_bf16" : "mfma_4x4x1";
}
template <int NACC, int K, int W, int OP>
static void mix_line(int iters, float* out) {
  const double n = iters * 16.0;
  const float m = timeit([&] { k_mix<NACC, K, W, OP, 0><<<256, 256 * W>>>(iters, out); });
  const float f = timeit([&] { k_mix<NACC, K, W, OP, 1><<<256, 256 * W>>>(iters, out); });
  const float b = timeit([&] { k_mix<NACC, K, W, OP, 2><<<256, 256 * W>>>(iters, out); });
  printf("  W=%d acc=%d K=%2d %-14s per MFMA slot: mfma %6.2f ns  fillers %6.2f ns  both %6.2f ns   both/sum %.2f  both/max %.2f\n", W, NACC, K,
         opname(OP), m * 1e6 / n, f * 1e6 / n, b * 1e6 / n, b / (m + f), b / (m > f ? m : f));
}
template <int NACC, int K, int OP>
static void pair_line(int iters, float* out) {
  const double n = iters * 16.0;
  const float m = timeit([&] { k_pair<NACC, K, OP><<<256, 512>>>(0, iters, out); });
  const float f = timeit([&] { k_pair<NACC, K, OP><<<256, 512>>>(1, iters, out); });
  const float b = timeit([&] { k_pair<NACC, K, OP><<<256, 512>>>(2, iters, out); });
  printf("  pair acc=%d K=%2d %-14s per MFMA slot: mfma wave %6.2f ns  valu wave %6.2f ns  both %6.2f ns   both/sum %.2f  both/max %.2f\n", NACC, K,
         opname(OP), m * 1e6 / n, f * 1e6 / n, b * 1e6 / n, b / (m + f), b / (m > f ? m : f));
}

template <int NACC, int K, int OP, int NOPS, int PM, int PV>
static void pair2_line(int iters, float* out) {
  const double n = iters * 16.0;
  const float m = timeit([&] { k_pair2<NACC, K, OP, NOPS, PM, PV><<<256, 512>>>(0, iters, out); });
  const float f = timeit([&] { k_pair2<NACC, K, OP, NOPS, PM, PV><<<256, 512>>>(1, iters, out); });
  const float b = timeit([&] { k_pair2<NACC, K, OP, NOPS, PM, PV><<<256, 512>>>(2, iters, out); });
  printf("  pair acc=%d K=%2d %-14s nops %2d prio mfma %d valu %d: mfma wave %6.2f ns  valu wave %6.2f ns  both %6.2f ns   both/sum %.2f  both/max %.2f\n",
         NACC, K, opname(OP), NOPS, PM, PV, m * 1e6 / n, f * 1e6 / n, b * 1e6 / n, b / (m + f), b / (m > f ? m : f));
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 1024 * 4);
  const int iters = 4000;
  printf("(d) one stream: [MFMA 32x32x16 bf16 into accumulator u mod acc, then K fillers] (13.3 ns = 32 cycles at 2.4 GHz)\n");
  mix_line<1, 8, 1, OP_FMA>(iters, out);
  mix_line<2, 8, 1, OP_FMA>(iters, out);
  mix_line<4, 8, 1, OP_FMA>(iters, out);
  mix_line<1, 12, 1, OP_FMA>(iters, out);
  mix_line<2, 12, 1, OP_FMA>(iters, out);
  mix_line<4, 12, 1, OP_FMA>(iters, out);
  mix_line<2, 4, 1, OP_EXP>(iters, out);
  mix_line<4, 4, 1, OP_EXP>(iters, out);
  mix_line<2, 6, 1, OP_PK>(iters, out);
  mix_line<2, 6, 1, OP_CVT>(iters, out);
  mix_line<1, 3, 1, OP_MFMA4>(iters, out);
  mix_line<2, 3, 1, OP_MFMA4>(iters, out);
  mix_line<1, 8, 2, OP_FMA>(iters, out);
  mix_line<2, 8, 2, OP_FMA>(iters, out);
  mix_line<4, 8, 2, OP_FMA>(iters, out);
  mix_line<2, 12, 2, OP_FMA>(iters, out);
  mix_line<2, 4, 2, OP_EXP>(iters, out);
  mix_line<2, 3, 2, OP_MFMA4>(iters, out);
  printf("(e) two waves per SIMD: an MFMA-only wave (acc accumulators) beside a filler-only wave (K per MFMA slot)\n");
  pair_line<1, 8, OP_FMA>(iters, out);
  pair_line<2, 8, OP_FMA>(iters, out);
  pair_line<4, 8, OP_FMA>(iters, out);
  pair_line<2, 14, OP_FMA>(iters, out);
  pair_line<1, 4, OP_EXP>(iters, out);
  pair_line<2, 4, OP_EXP>(iters, out);
  pair_line<2, 6, OP_PK>(iters, out);
  pair_line<1, 3, OP_MFMA4>(iters, out);
  pair_line<2, 3, OP_MFMA4>(iters, out);
  printf("(f) the pair with s_nop cycles behind every MFMA / with priorities\n");
  pair2_line<1, 8, OP_FMA, 8, 0, 0>(iters, out);
  pair2_line<1, 8, OP_FMA, 16, 0, 0>(iters, out);
  pair2_line<1, 8, OP_FMA, 24, 0, 0>(iters, out);
  pair2_line<1, 8, OP_FMA, 28, 0, 0>(iters, out);
  pair2_line<1, 8, OP_FMA, 32, 0, 0>(iters, out);
  pair2_line<2, 8, OP_FMA, 24, 0, 0>(iters, out);
  pair2_line<1, 12, OP_FMA, 24, 0, 0>(iters, out);
  pair2_line<1, 4, OP_EXP, 24, 0, 0>(iters, out);
  pair2_line<1, 6, OP_PK, 24, 0, 0>(iters, out);
  pair2_line<1, 3, OP_MFMA4, 24, 0, 0>(iters, out);
  pair2_line<1, 8, OP_FMA, 0, 0, 1>(iters, out);
  pair2_line<1, 8, OP_FMA, 0, 1, 0>(iters, out);
  pair2_line<1, 8, OP_FMA, 0, 0, 3>(iters, out);
  pair2_line<2, 8, OP_FMA, 0, 0, 1>(iters, out);
  pair2_line<1, 8, OP_FMA, 24, 0, 1>(iters, out);
  pair2_line<1, 8, OP_FMA, 24, 1, 0>(iters, out);
  printf("(g) the pair at priority (MFMA wave 0, filler wave 1), by filler kind\n");
  pair2_line<1, 8, OP_FMA, 0, 0, 1>(iters, out);
  pair2_line<1, 4, OP_EXP, 0, 0, 1>(iters, out);
  pair2_line<1, 6, OP_PK, 0, 0, 1>(iters, out);
  pair2_line<1, 6, OP_CVT, 0, 0, 1>(iters, out);
  pair2_line<1, 3, OP_MFMA4, 0, 0, 1>(iters, out);
  pair2_line<1, 2, OP_MFMA4, 0, 0, 1>(iters, out);
  pair2_line<2, 3, OP_MFMA4, 0, 0, 1>(iters, out);
  return 0;
}
