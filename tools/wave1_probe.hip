// What ONE wave per SIMD can issue (gfx950): cycles per instruction of inline-asm streams, by encoding size and dependency
// distance, with 1 and 2 waves per SIMD; then a bf16 MFMA with K fillers of each kind behind it.
//   hipcc -O3 --offload-arch=gfx950 tools/wave1_probe.hip -o /tmp/wave1_probe && /tmp/wave1_probe
// Cycles are s_memtime deltas of wave 0 of block 0 over `iters` iterations of an unrolled body of 64 instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

enum { ADD32 = 0, ADD64, FMA, EXP, RCP, CVT, ANDLIT, ANDSGPR, LSHL, SUB32, MOV, PERM, MUL32, FMAC };
static const char* NAMES[] = {"v_add_f32 e32", "v_add_f32 e64", "v_fma_f32 (e64)", "v_exp_f32", "v_rcp_f32", "v_cvt_pk_bf16_f32 (e64)",
                              "v_and_b32 literal (8 B)", "v_and_b32 sgpr (4 B)", "v_lshlrev_b32 16 (4 B)", "v_sub_f32 e32", "v_mov_b32",
                              "v_perm_b32 (e64)", "v_mul_f32 e32", "v_fmac_f32 e32"};

template <int OP>
__device__ __forceinline__ void op(float& d, float a, float b, unsigned m) {
  if (OP == ADD32) asm volatile("v_add_f32_e32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  if (OP == ADD64) asm volatile("v_add_f32_e64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  if (OP == FMA) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(a));
  if (OP == EXP) asm volatile("v_exp_f32_e32 %0, %1" : "=v"(d) : "v"(a));
  if (OP == RCP) asm volatile("v_rcp_f32_e32 %0, %1" : "=v"(d) : "v"(a));
  if (OP == CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  if (OP == ANDLIT) asm volatile("v_and_b32_e32 %0, 0xffff0000, %1" : "=v"(d) : "v"(a));
  if (OP == ANDSGPR) asm volatile("v_and_b32_e32 %0, %2, %1" : "=v"(d) : "v"(a), "s"(m));
  if (OP == LSHL) asm volatile("v_lshlrev_b32_e32 %0, 16, %1" : "=v"(d) : "v"(a));
  if (OP == SUB32) asm volatile("v_sub_f32_e32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  if (OP == MOV) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(d) : "v"(a));
  if (OP == PERM) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(m));
  if (OP == MUL32) asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  if (OP == FMAC) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(d) : "v"(a), "v"(b));
}

// DIST = dependency distance: instruction i writes register i % DIST and reads what instruction i - DIST wrote
template <int OP, int DIST, int W>
__global__ void __launch_bounds__(256 * W, W) k_stream(int iters, float* out, unsigned long long* cyc) {
  float v[16], b = 1.0001f;
  unsigned m = 0xffff0000u + (threadIdx.x >> 10);
  for (int r = 0; r < 16; ++r) v[r] = threadIdx.x * 1e-3f + r;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 64; ++u) op<OP>(v[u % DIST], v[u % DIST], b, m);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int r = 0; r < 16; ++r) s += v[r];
  out[blockIdx.x * 256 * W + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}

// one bf16 MFMA (dependent chain into one accumulator) + K fillers of kind OP (distance 8), 16 slots per iteration
template <int OP, int K, int W>
__global__ void __launch_bounds__(256 * W, W) k_gap(int iters, float* out, unsigned long long* cyc) {
  u32x4 a = {threadIdx.x * 3u + 0x3f803f80u, 0x3f803f80u, 0x3f003f00u, 0x3e803e80u}, bb = {0x3f803f80u, 0x3f003f00u, 0x3f803f80u, 0x3f003f00u};
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0;
  float v[16], b = 1.0001f;
  unsigned m = 0xffff0000u + (threadIdx.x >> 10);
  for (int r = 0; r < 16; ++r) v[r] = threadIdx.x * 1e-3f + r;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(bb));
#pragma unroll
      for (int k = 0; k < K; ++k) op<OP>(v[(u * K + k) % 8], v[(u * K + k) % 8], b, m);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int r = 0; r < 16; ++r) s += v[r] + c[r];
  out[blockIdx.x * 256 * W + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}

// (c) two waves per SIMD: waves 0-3 run a dependent chain of bf16 MFMAs with KN fillers of kind FOP (v_nop when FOP < 0)
// behind each, waves 4-7 a stream of VOP instructions of kind OP (distance 8), VPER per MFMA slot.  mode 0 / 1 / 2 =
// MFMA waves only / VALU waves only / both; wall time by events per MFMA slot.
template <int KN, int OP, int VPER>
__global__ void __launch_bounds__(512, 2) k_pair(int mode, int iters, float* out) {
  const int wave = threadIdx.x >> 6;
  float v[16], b = 1.0001f;
  unsigned m = 0xffff0000u + (threadIdx.x >> 10);
  for (int r = 0; r < 16; ++r) v[r] = threadIdx.x * 1e-3f + r;
  if (wave < 4) {
    if (mode == 1) return;
    u32x4 a = {threadIdx.x * 3u + 0x3f803f80u, 0x3f803f80u, 0x3f003f00u, 0x3e803e80u}, bb = {0x3f803f80u, 0x3f003f00u, 0x3f803f80u, 0x3f003f00u};
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(bb));
#pragma unroll
        for (int k = 0; k < KN; ++k) asm volatile("v_nop");
      }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c[r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  } else {
    if (mode == 0) return;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16 * VPER; ++u) op<OP>(v[u % 8], v[u % 8], b, m);
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += v[r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  }
}
template <int KN, int OP, int VPER>
static void pline(int iters, float* out) {
  auto timeit = [&](int mode) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k_pair<KN, OP, VPER><<<256, 512>>>(mode, iters, out);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0); k_pair<KN, OP, VPER><<<256, 512>>>(mode, iters, out); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    return best * 1e6 / (iters * 16.0);
  };
  const double m0 = timeit(0), v0 = timeit(1), b0 = timeit(2);
  printf("  %d v_nop behind each MFMA, partner %2d x %-26s per MFMA slot: mfma waves %6.2f ns  valu waves %6.2f ns  both %6.2f ns   both/sum %.2f  both/max %.2f\n",
         KN, VPER, NAMES[OP], m0, v0, b0, b0 / (m0 + v0), b0 / (m0 > v0 ? m0 : v0));
}

// (d) wall time (events) of W waves per SIMD each running a stream of NA x OPA then NB x OPB (all independent, distance 8
// per kind), per instruction of one wave's stream
template <int OPA, int NA, int OPB, int NB, int W>
__global__ void __launch_bounds__(256 * W, W) k_mix2(int iters, float* out) {
  float v[16], w[16], b = 1.0001f;
  unsigned m = 0xffff0000u + (threadIdx.x >> 10);
  for (int r = 0; r < 16; ++r) { v[r] = threadIdx.x * 1e-3f + r; w[r] = v[r] + 0.5f; }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int k = 0; k < NA; ++k) op<OPA>(v[(u * NA + k) % 8], v[(u * NA + k) % 8], b, m);
#pragma unroll
      for (int k = 0; k < NB; ++k) op<OPB>(w[(u * NB + k) % 8], w[(u * NB + k) % 8], b, m);
    }
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += v[r] + w[r];
  out[blockIdx.x * 256 * W + threadIdx.x] = s;
}
template <int OPA, int NA, int OPB, int NB, int W>
static double mix2(int iters, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k_mix2<OPA, NA, OPB, NB, W><<<256, 256 * W>>>(iters, out);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0); k_mix2<OPA, NA, OPB, NB, W><<<256, 256 * W>>>(iters, out); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  return best * 1e6 / (iters * 8.0 * (NA + NB));  // ns per instruction of one wave's stream
}
template <int OPA, int NA, int OPB, int NB>
static void mline(float* out) {
  printf("  %d x %-24s + %d x %-24s ns per instruction of one stream: 1 wave/SIMD %6.3f   2 waves/SIMD %6.3f   4 waves/SIMD %6.3f\n", NA, NAMES[OPA], NB,
         NAMES[OPB], mix2<OPA, NA, OPB, NB, 1>(4000, out), mix2<OPA, NA, OPB, NB, 2>(4000, out), mix2<OPA, NA, OPB, NB, 4>(2000, out));
}

static float* g_out;
static unsigned long long* g_cyc;
template <int OP, int DIST, int W>
static double stream(int iters) {
  k_stream<OP, DIST, W><<<256, 256 * W>>>(iters, g_out, g_cyc);
  k_stream<OP, DIST, W><<<256, 256 * W>>>(iters, g_out, g_cyc);
  hipDeviceSynchronize();
  unsigned long long c;
  hipMemcpy(&c, g_cyc, 8, hipMemcpyDeviceToHost);
  return (double)c / (iters * 64.0);
}
template <int OP, int K, int W>
static double gap(int iters) {
  k_gap<OP, K, W><<<256, 256 * W>>>(iters, g_out, g_cyc);
  k_gap<OP, K, W><<<256, 256 * W>>>(iters, g_out, g_cyc);
  hipDeviceSynchronize();
  unsigned long long c;
  hipMemcpy(&c, g_cyc, 8, hipMemcpyDeviceToHost);
  return (double)c / (iters * 16.0);
}
template <int OP>
static void line() {
  const int it = 2000;
  printf("  %-28s 1 wave/SIMD: dist 1 %6.2f  dist 2 %6.2f  dist 4 %6.2f  dist 8 %6.2f  dist 16 %6.2f   | 2 waves/SIMD (per instruction of one stream): dist 1 %6.2f  dist 8 %6.2f\n",
         NAMES[OP], stream<OP, 1, 1>(it), stream<OP, 2, 1>(it), stream<OP, 4, 1>(it), stream<OP, 8, 1>(it), stream<OP, 16, 1>(it),
         stream<OP, 1, 2>(it), stream<OP, 8, 2>(it));
}
template <int OP>
static void gline() {
  const int it = 2000;
  printf("  %-28s cycles per [MFMA + K fillers], 1 wave/SIMD: K=0 %6.1f  K=2 %6.1f  K=4 %6.1f  K=6 %6.1f  K=8 %6.1f  K=12 %6.1f  K=16 %6.1f | 2 waves/SIMD: K=4 %6.1f K=8 %6.1f\n",
         NAMES[OP], gap<OP, 0, 1>(it), gap<OP, 2, 1>(it), gap<OP, 4, 1>(it), gap<OP, 6, 1>(it), gap<OP, 8, 1>(it), gap<OP, 12, 1>(it),
         gap<OP, 16, 1>(it), gap<OP, 4, 2>(it), gap<OP, 8, 2>(it));
}
int main() {
  hipMalloc(&g_out, 256 * 512 * 4);
  hipMalloc(&g_cyc, 8);
  printf("(a) cycles per instruction of a stream, by dependency distance (s_memtime ticks)\n");
  line<ADD32>(); line<ADD64>(); line<SUB32>(); line<MUL32>(); line<FMAC>(); line<FMA>(); line<MOV>(); line<LSHL>(); line<ANDSGPR>(); line<ANDLIT>();
  line<PERM>(); line<CVT>(); line<EXP>(); line<RCP>();
  printf("(b) a dependent chain of v_mfma_f32_32x32x16_bf16 with K independent fillers (distance 8) behind each\n");
  gline<ADD32>(); gline<ADD64>(); gline<FMA>(); gline<LSHL>(); gline<ANDLIT>(); gline<CVT>(); gline<EXP>();
  printf("(c) two waves per SIMD: an MFMA chain with K v_nop behind each MFMA beside a VALU-only partner (13.3 ns = 32 cycles at 2.4 GHz)\n");
  pline<0, ADD32, 6>(2000, g_out); pline<2, ADD32, 6>(2000, g_out); pline<3, ADD32, 6>(2000, g_out); pline<4, ADD32, 6>(2000, g_out);
  pline<5, ADD32, 6>(2000, g_out); pline<6, ADD32, 6>(2000, g_out);
  pline<0, ADD32, 4>(2000, g_out); pline<4, ADD32, 4>(2000, g_out); pline<5, ADD32, 4>(2000, g_out);
  pline<0, FMA, 6>(2000, g_out); pline<4, FMA, 6>(2000, g_out); pline<5, FMA, 6>(2000, g_out);
  pline<0, EXP, 3>(2000, g_out); pline<4, EXP, 3>(2000, g_out); pline<5, EXP, 3>(2000, g_out);
  pline<0, CVT, 6>(2000, g_out); pline<4, CVT, 6>(2000, g_out);
  printf("(d) W waves per SIMD, each the same stream; wall time per instruction of one stream (0.417 ns = 1 cycle at 2.4 GHz)\n");
  mline<ADD32, 8, ADD32, 0>(g_out); mline<FMA, 8, FMA, 0>(g_out); mline<EXP, 8, EXP, 0>(g_out); mline<CVT, 8, CVT, 0>(g_out);
  mline<EXP, 1, ADD32, 1>(g_out); mline<EXP, 1, ADD32, 2>(g_out); mline<EXP, 1, ADD32, 3>(g_out); mline<EXP, 2, ADD32, 2>(g_out);
  mline<EXP, 1, FMA, 2>(g_out);
  return 0;
}
