// Issue-cost probe for the f32 ALU of gfx950 (MI355X): what the fused trajectory kernel's cycle budget rests on.
//   (a) v_fma_f32 / v_exp_f32 / v_pk_fma_f32 issue cost with 1 and 2 waves per SIMD
//   (b) ONE wave: k independent VALU (or transcendental) instructions placed after every v_mfma_f32_32x32x2_f32 of a
//       dependent chain -- do any of them hide under the MFMA's 64 cycles?
//   (c) TWO waves on a SIMD: one in an MFMA chain, the partner in a VALU stream -- do the times add?
//   (d) v_mfma_f32_4x4x1_16b_f32: dependent-chain cost with 1 / 2 / 4 accumulators
// Cycles are s_memtime deltas of wave 0 of block 0 (shader clock), one 256-thread (or 512-thread) block per CU.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/issue_probe tools/issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define STAMP() __builtin_amdgcn_s_memtime()

// Elapsed cycles of the SLOWEST wave of block 0: with two waves per SIMD the arbiter favours the older one, whose own
// stamps would look like a stand-alone run.
__device__ __forceinline__ void report(unsigned long long t0, unsigned long long t1, unsigned long long* cyc) {
  __shared__ unsigned long long t_wave[16];
  if ((threadIdx.x & 63) == 0) t_wave[threadIdx.x >> 6] = t1 - t0;
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    unsigned long long m = 0;
    for (unsigned w = 0; w < blockDim.x / 64; ++w) m = t_wave[w] > m ? t_wave[w] : m;
    *cyc = m;
  }
}

enum { OP_FMA = 0, OP_EXP = 1, OP_PKFMA = 2 };

template <int OP>
__device__ __forceinline__ void valu8(float (&v)[8], float b) {
  if (OP == OP_FMA) {
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = __builtin_fmaf(v[r], b, 0.5f);
  } else if (OP == OP_EXP) {
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = __builtin_amdgcn_exp2f(v[r]);
  } else {
#pragma unroll
    for (int r = 0; r < 8; r += 2) {
      f32x2 x = {v[r], v[r + 1]};
      const f32x2 bb = {b, b}, cc = {0.5f, 0.25f};
      x = __builtin_elementwise_fma(x, bb, cc);
      v[r] = x[0];
      v[r + 1] = x[1];
    }
  }
}

// (a) W waves per SIMD, each a stream of independent VALU instructions of one kind
template <int W, int OP>
__global__ void __launch_bounds__(256 * W, W) k_valu(int iters, float* out, unsigned long long* cyc) {
  float b = 1.0001f, v[8];
  for (int r = 0; r < 8; ++r) v[r] = threadIdx.x * 1e-3f + r;
  const unsigned long long t0 = STAMP();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) valu8<OP>(v, b);
  }
  const unsigned long long t1 = STAMP();
  float s = 0;
  for (int r = 0; r < 8; ++r) s += v[r];
  out[blockIdx.x * 256 * W + threadIdx.x] = s;
  report(t0, t1, cyc);
}

// (b) one wave per SIMD (W = 1) or two (W = 2): 16 dependent MFMAs per iteration, K fillers after each
template <int W, int K, int OP>
__global__ void __launch_bounds__(256 * W, W) k_mix(int iters, float* out, unsigned long long* cyc) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f, v[8];
  for (int r = 0; r < 8; ++r) v[r] = a + r;
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0;
  const unsigned long long t0 = STAMP();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (OP == OP_EXP) v[k & 7] = __builtin_amdgcn_exp2f(v[k & 7]);
        else v[k & 7] = __builtin_fmaf(v[k & 7], b, 0.5f);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = STAMP();
  float s = 0;
  for (int r = 0; r < 16; ++r) s += c[r];
  for (int r = 0; r < 8; ++r) s += v[r];
  out[blockIdx.x * 256 * W + threadIdx.x] = s;
  report(t0, t1, cyc);
}

// (c) waves 0-3 (one per SIMD): MFMA chain of n_mfma; waves 4-7 (their partners): n_valu VALU instructions.
// mode 0: MFMA waves only, 1: VALU waves only, 2: both.  The block's time = the stamp difference seen by wave 0 / 4.
template <int OP>
__global__ void __launch_bounds__(512, 2) k_pair(int mode, int iters, float* out, unsigned long long* cyc) {
  const int wave = threadIdx.x >> 6;
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  __shared__ unsigned long long t_end[8];
  __syncthreads();
  const unsigned long long t0 = STAMP();
  float s = 0;
  if (wave < 4) {
    if (mode != 1) {
      f32x16 c;
      for (int r = 0; r < 16; ++r) c[r] = 0;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
      }
      for (int r = 0; r < 16; ++r) s += c[r];
    }
  } else {
    if (mode != 0) {
      float v[8];
      for (int r = 0; r < 8; ++r) v[r] = a + r;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 32; ++u) valu8<OP>(v, b);  // 256 instructions per iteration (16 MFMAs = 1024 cycles)
      }
      for (int r = 0; r < 8; ++r) s += v[r];
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
  const unsigned long long t1 = STAMP();
  if ((threadIdx.x & 63) == 0) t_end[wave] = t1 - t0;
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    unsigned long long m = 0;
    for (int w = 0; w < 8; ++w) m = t_end[w] > m ? t_end[w] : m;
    *cyc = m;
  }
}

// (d) 4x4x1 chains
template <int W, int NACC>
__global__ void __launch_bounds__(256 * W, W) k44(int iters, float* out, unsigned long long* cyc) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  f32x4 d[4];
  for (int i = 0; i < 4; ++i) d[i] = (f32x4){0, 0, 0, 0};
  const unsigned long long t0 = STAMP();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 64; ++u) d[u % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d[u % NACC], 0, 0, 0);
  }
  const unsigned long long t1 = STAMP();
  out[blockIdx.x * 256 * W + threadIdx.x] = d[0][0] + d[1][0] + d[2][0] + d[3][0];
  report(t0, t1, cyc);
}

// (e) 64 4x4x1 MFMAs (4 accumulators) and 64 independent v_fma_f32 per iteration, alternating in groups of G:
// G = 1 is MFMA, FMA, MFMA, FMA, ...; G = 64 is all MFMAs then all FMAs.  What does a switch between the two cost?
template <int W, int G, int BIG>
__global__ void __launch_bounds__(256 * W, W) k_alt(int iters, float* out, unsigned long long* cyc) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  f32x4 d[4];
  f32x16 D[2];
  for (int i = 0; i < 4; ++i) d[i] = (f32x4){0, 0, 0, 0};
  for (int i = 0; i < 2; ++i)
    for (int r = 0; r < 16; ++r) D[i][r] = 0.0f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = a + i;
  const unsigned long long t0 = STAMP();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u0 = 0; u0 < 64; u0 += G) {
#pragma unroll
      for (int u = u0; u < u0 + G; ++u) {
        if (BIG) D[u & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, D[u & 1], 0, 0, 0);
        else d[u & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d[u & 3], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = u0; u < u0 + G; ++u) v[u & 7] = __builtin_fmaf(v[u & 7], b, a);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = STAMP();
  float s = 0.0f;
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 256 * W + threadIdx.x] = d[0][0] + d[1][0] + d[2][0] + d[3][0] + D[0][0] + D[1][0] + s;
  report(t0, t1, cyc);
}

static float* g_out;
static unsigned long long* g_cyc;

template <typename F>
static double run(F launch) {
  launch();  // warm
  hipDeviceSynchronize();
  launch();
  hipDeviceSynchronize();
  unsigned long long c = 0;
  hipMemcpy(&c, g_cyc, 8, hipMemcpyDeviceToHost);
  return (double)c;
}

int main() {
  hipMalloc(&g_out, 256 * 512 * sizeof(float));
  hipMalloc(&g_cyc, 8);
  const int it = 2000;
  printf("(a) cycles per instruction of one wave's stream (128 instr per iteration), all CUs busy; with 2 waves per SIMD\n"
         "    the time until BOTH have finished their stream, per instruction of one stream\n");
  printf("  v_fma_f32    1 wave/SIMD %.2f   2 waves/SIMD %.2f\n",
         run([&] { hipLaunchKernelGGL((k_valu<1, OP_FMA>), dim3(256), dim3(256), 0, 0, it, g_out, g_cyc); }) / it / 128,
         run([&] { hipLaunchKernelGGL((k_valu<2, OP_FMA>), dim3(256), dim3(512), 0, 0, it, g_out, g_cyc); }) / it / 128);
  printf("  v_exp_f32    1 wave/SIMD %.2f   2 waves/SIMD %.2f\n",
         run([&] { hipLaunchKernelGGL((k_valu<1, OP_EXP>), dim3(256), dim3(256), 0, 0, it, g_out, g_cyc); }) / it / 128,
         run([&] { hipLaunchKernelGGL((k_valu<2, OP_EXP>), dim3(256), dim3(512), 0, 0, it, g_out, g_cyc); }) / it / 128);
  printf("  v_pk_fma_f32 1 wave/SIMD %.2f   2 waves/SIMD %.2f (64 instr per iteration)\n",
         run([&] { hipLaunchKernelGGL((k_valu<1, OP_PKFMA>), dim3(256), dim3(256), 0, 0, it, g_out, g_cyc); }) / it / 64,
         run([&] { hipLaunchKernelGGL((k_valu<2, OP_PKFMA>), dim3(256), dim3(512), 0, 0, it, g_out, g_cyc); }) / it / 64);
  printf("(b) cycles per [v_mfma_f32_32x32x2_f32 + K fillers], dependent chain, per wave\n");
#define MIX(W, K, OP, name) \
  printf("  W=%d K=%2d %s: %.1f\n", W, K, name, \
         run([&] { hipLaunchKernelGGL((k_mix<W, K, OP>), dim3(256), dim3(256 * W), 0, 0, it, g_out, g_cyc); }) / it / 16)
  MIX(1, 0, OP_FMA, "fma"); MIX(1, 2, OP_FMA, "fma"); MIX(1, 4, OP_FMA, "fma"); MIX(1, 8, OP_FMA, "fma"); MIX(1, 16, OP_FMA, "fma");
  MIX(1, 2, OP_EXP, "exp"); MIX(1, 4, OP_EXP, "exp"); MIX(1, 8, OP_EXP, "exp");
  MIX(2, 0, OP_FMA, "fma"); MIX(2, 4, OP_FMA, "fma"); MIX(2, 8, OP_FMA, "fma"); MIX(2, 16, OP_FMA, "fma");
  MIX(2, 4, OP_EXP, "exp"); MIX(2, 8, OP_EXP, "exp");
  printf("(c) two waves per SIMD, cycles per iteration: MFMA wave alone (16 MFMAs), VALU wave alone (256 instr), both\n");
  for (int op = 0; op < 2; ++op) {
    double r[3];
    for (int mode = 0; mode < 3; ++mode)
      r[mode] = op == 0
                    ? run([&] { hipLaunchKernelGGL((k_pair<OP_FMA>), dim3(256), dim3(512), 0, 0, mode, it, g_out, g_cyc); }) / it
                    : run([&] { hipLaunchKernelGGL((k_pair<OP_EXP>), dim3(256), dim3(512), 0, 0, mode, it, g_out, g_cyc); }) / it;
    printf("  %s: mfma %.0f  valu %.0f  both %.0f\n", op == 0 ? "v_fma" : "v_exp", r[0], r[1], r[2]);
  }
  printf("(d) cycles per v_mfma_f32_4x4x1_16b_f32, per wave\n");
#define K44(W, N) \
  printf("  W=%d accumulators=%d: %.1f\n", W, N, \
         run([&] { hipLaunchKernelGGL((k44<W, N>), dim3(256), dim3(256 * W), 0, 0, it, g_out, g_cyc); }) / it / 64)
  K44(1, 1); K44(1, 2); K44(1, 4); K44(2, 1); K44(2, 2); K44(2, 4);
  printf("(e) cycles per iteration of 64 MFMAs + 64 independent v_fma_f32 alternating in groups of G (4x4x1 alone 64 x 8.4 =\n"
         "    538, 32x32x2 alone 4096, the 64 FMAs alone 141)\n");
#define ALT(W, G, BIG) \
  printf("  %s W=%d G=%2d: %.0f\n", BIG ? "32x32x2" : "4x4x1  ", W, G, \
         run([&] { hipLaunchKernelGGL((k_alt<W, G, BIG>), dim3(256), dim3(256 * W), 0, 0, it, g_out, g_cyc); }) / it)
  ALT(1, 1, 0); ALT(1, 2, 0); ALT(1, 4, 0); ALT(1, 16, 0); ALT(1, 64, 0);
  ALT(2, 1, 0); ALT(2, 4, 0); ALT(2, 64, 0);
  ALT(1, 1, 1); ALT(1, 4, 1); ALT(1, 64, 1);
  ALT(2, 1, 1); ALT(2, 64, 1);
  return 0;
}
