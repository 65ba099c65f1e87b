// Does vector-ALU work run beside v_mfma_f32_32x32x16_bf16 on one SIMD?  (run on the GPU box)
//   hipcc -O3 --offload-arch=gfx950 tools/bf16_coexec_probe.hip -o tools/bf16_coexec_probe && tools/bf16_coexec_probe
// (a) issue cost of the instructions the operand split uses, one and two waves per SIMD;
// (b) two waves per SIMD: an MFMA-only wave (dependent chain), a VALU-only wave, both together;
// (c) one wave: an MFMA chain with K independent VALU instructions between consecutive MFMAs.
// Cycles are derived from the wall time at the clock a bare v_fma loop implies (2.22 cycles per v_fma_f32, one wave).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

enum { OP_FMA = 0, OP_EXP = 1, OP_CVT = 2, OP_PKADD = 3, OP_AND = 4, OP_SUB = 5 };

template <int OP>
__device__ __forceinline__ void valu_block(float (&v)[8], float b) {
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    if (OP == OP_FMA) v[r] = __builtin_fmaf(v[r], b, 0.5f);
    if (OP == OP_EXP) v[r] = __builtin_amdgcn_exp2f(v[r]);
    if (OP == OP_SUB) v[r] = v[r] - b;
    if (OP == OP_AND) v[r] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v[r]) & 0xffff0000u);
  }
  if (OP == OP_CVT) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      unsigned o;
      asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(o) : "v"(v[r]), "v"(v[(r + 1) & 7]));
      v[r] = __builtin_bit_cast(float, o);
    }
  }
  if (OP == OP_PKADD) {
#pragma unroll
    for (int r = 0; r < 8; r += 2) {
      f32x2 a = {v[r], v[r + 1]};
      asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(a));
      v[r] = a[0];
      v[r + 1] = a[1];
    }
  }
}

// W waves per SIMD, every wave the same VALU stream: 8 (4 for pk_add) instructions x 16 per iteration
template <int OP, int W>
__global__ void __launch_bounds__(256 * W, W) k_valu(int iters, float* out) {
  float v[8];
  for (int r = 0; r < 8; ++r) v[r] = threadIdx.x * 1e-3f + r;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) valu_block<OP>(v, 1.0001f);
  }
  float s = 0;
  for (int r = 0; r < 8; ++r) s += v[r];
  out[blockIdx.x * 256 * W + threadIdx.x] = s;
}

__device__ __forceinline__ f32x16 mm(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// waves 0-3: dependent chain of 16 bf16 MFMAs per iteration; waves 4-7: 128 VALU instructions per iteration
// mode 0: MFMA waves only, 1: VALU waves only, 2: both
__device__ unsigned long long g_stamp[256 * 8 * 2];  // per wave: shader cycles, 100 MHz ticks
template <int OP>
__global__ void __launch_bounds__(512, 2) k_pair(int mode, int iters, float* out) {
  const int wave = threadIdx.x >> 6;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  struct Stamp {
    unsigned long long c0, r0; int slot;
    __device__ ~Stamp() {
      if ((threadIdx.x & 63) == 0) {
        g_stamp[2 * slot] = __builtin_amdgcn_s_memtime() - c0;
        g_stamp[2 * slot + 1] = __builtin_amdgcn_s_memrealtime() - r0;
      }
    }
  } stamp{c0, r0, (int)(blockIdx.x * 8 + wave)};
  if (wave < 4) {
    if (mode == 1) return;
    u32x4 a = {threadIdx.x * 3u + 0x3f803f80u, 0x3f803f80u, 0x3f003f00u, 0x3e803e80u}, b = {0x3f803f80u, 0x3f003f00u, 0x3f803f80u, 0x3f003f00u};
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) c = mm(a, b, c);
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c[r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  } else {
    if (mode == 0) return;
    float v[8];
    for (int r = 0; r < 8; ++r) v[r] = threadIdx.x * 1e-3f + r;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) valu_block<OP>(v, 1.0001f);
    }
    float s = 0;
    for (int r = 0; r < 8; ++r) s += v[r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  }
}

// one stream: [MFMA + K independent v_fma] x 16 per iteration, W waves per SIMD
template <int K, int W, int OP>
__global__ void __launch_bounds__(256 * W, W) k_mix(int iters, float* out) {
  u32x4 a = {threadIdx.x * 3u + 0x3f803f80u, 0x3f803f80u, 0x3f003f00u, 0x3e803e80u}, b = {0x3f803f80u, 0x3f003f00u, 0x3f803f80u, 0x3f003f00u};
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0;
  float v[16];
  for (int r = 0; r < 16; ++r) v[r] = threadIdx.x * 1e-3f + r;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      c = mm(a, b, c);
#pragma unroll
      for (int r = 0; r < K; ++r) {
        if (OP == OP_FMA) v[r] = __builtin_fmaf(v[r], 1.0001f, 0.5f);
        else v[r] = __builtin_amdgcn_exp2f(v[r]);
      }
    }
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += c[r] + v[r];
  out[blockIdx.x * 256 * W + threadIdx.x] = s;
}

template <typename F>
static float timeit(F f) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  f(); f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

static double g_ghz = 2.4;
static double cyc(float ms, double n) { return ms * 1e-3 * g_ghz * 1e9 / n; }

template <int OP>
static void valu_line(const char* name, int iters, float* out, int per_iter) {
  const float t1 = timeit([&] { k_valu<OP, 1><<<256, 256>>>(iters, out); });
  const float t2 = timeit([&] { k_valu<OP, 2><<<256, 512>>>(iters, out); });
  printf("  %-22s 1 wave/SIMD %6.2f   2 waves/SIMD %6.2f per SIMD-instruction %6.2f\n", name,
         cyc(t1, (double)iters * per_iter), cyc(t2, (double)iters * per_iter), cyc(t2, (double)iters * per_iter) / 2);
}
template <int OP>
static void pair_line(const char* name, int iters, float* out) {
  const float a = timeit([&] { k_pair<OP><<<256, 512>>>(0, iters, out); });
  const float b = timeit([&] { k_pair<OP><<<256, 512>>>(1, iters, out); });
  const float c = timeit([&] { k_pair<OP><<<256, 512>>>(2, iters, out); });
  printf("  %-22s mfma wave %7.0f   valu wave %7.0f   both %7.0f   (sum %7.0f)\n", name, cyc(a, iters), cyc(b, iters),
         cyc(c, iters), cyc(a, iters) + cyc(b, iters));
  // the same three runs by the waves' own clocks: shader cycles per iteration (s_memtime) and the clock they ran at
  for (int mode = 0; mode < 3; ++mode) {
    k_pair<OP><<<256, 512>>>(mode, iters, out);
    hipDeviceSynchronize();
    k_pair<OP><<<256, 512>>>(mode, iters, out);
    hipDeviceSynchronize();
    static unsigned long long h[256 * 8 * 2];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamp), sizeof(h));
    double cm = 0, cv = 0, rm = 0, rv = 0;
    for (int b = 0; b < 256; ++b)
      for (int w = 0; w < 8; ++w) {
        (w < 4 ? cm : cv) += (double)h[2 * (b * 8 + w)] / 1024;
        (w < 4 ? rm : rv) += (double)h[2 * (b * 8 + w) + 1] / 1024;
      }
    printf("      mode %d by s_memtime: mfma waves %7.0f cycles/iter at %.2f GHz   valu waves %7.0f cycles/iter at %.2f GHz\n",
           mode, cm / iters, rm > 0 ? cm / rm * 0.1 : 0.0, cv / iters, rv > 0 ? cv / rv * 0.1 : 0.0);
  }
}
template <int K, int W, int OP>
static void mix_line(int iters, float* out) {
  const float t = timeit([&] { k_mix<K, W, OP><<<256, 256 * W>>>(iters, out); });
  printf("  W=%d K=%2d %s: %6.1f\n", W, K, OP == OP_FMA ? "fma" : "exp", cyc(t, (double)iters * 16));
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 1024 * 4);
  const int iters = 4000;
  {  // clock from the bare fma loop: 2.22 cycles per instruction at one wave per SIMD (profiles/r02_issue_probe.txt)
    const float t = timeit([&] { k_valu<OP_FMA, 1><<<256, 256>>>(iters, out); });
    g_ghz = 2.22 * iters * 128.0 / (t * 1e-3) / 1e9;
    printf("clock implied by the v_fma_f32 loop: %.3f GHz\n", g_ghz);
  }
  printf("(a) cycles per instruction of one wave's stream\n");
  valu_line<OP_FMA>("v_fma_f32", iters, out, 128);
  valu_line<OP_EXP>("v_exp_f32", iters, out, 128);
  valu_line<OP_CVT>("v_cvt_pk_bf16_f32", iters, out, 128);
  valu_line<OP_PKADD>("v_pk_add_f32", iters, out, 64);
  valu_line<OP_AND>("v_and_b32 (literal)", iters, out, 128);
  valu_line<OP_SUB>("v_sub_f32", iters, out, 128);
  printf("(b) two waves per SIMD, cycles per iteration: 16 dependent v_mfma_f32_32x32x16_bf16 | 128 VALU (64 pk) instructions\n");
  pair_line<OP_FMA>("v_fma_f32", iters, out);
  pair_line<OP_EXP>("v_exp_f32", iters, out);
  pair_line<OP_CVT>("v_cvt_pk_bf16_f32", iters, out);
  pair_line<OP_PKADD>("v_pk_add_f32", iters, out);
  printf("(c) cycles per [v_mfma_f32_32x32x16_bf16 + K fillers], dependent MFMA chain, per wave\n");
  mix_line<0, 1, OP_FMA>(iters, out);
  mix_line<4, 1, OP_FMA>(iters, out);
  mix_line<8, 1, OP_FMA>(iters, out);
  mix_line<12, 1, OP_FMA>(iters, out);
  mix_line<16, 1, OP_FMA>(iters, out);
  mix_line<2, 1, OP_EXP>(iters, out);
  mix_line<4, 1, OP_EXP>(iters, out);
  mix_line<0, 2, OP_FMA>(iters, out);
  mix_line<4, 2, OP_FMA>(iters, out);
  mix_line<8, 2, OP_FMA>(iters, out);
  mix_line<16, 2, OP_FMA>(iters, out);
  mix_line<2, 2, OP_EXP>(iters, out);
  mix_line<4, 2, OP_EXP>(iters, out);
  return 0;
}
