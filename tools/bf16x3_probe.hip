// Probe for the bf16x3 formulation of the 32x32x32 f32 products of the fused trajectory kernel (run on the GPU box):
//   hipcc -O3 --offload-arch=gfx950 tools/bf16x3_probe.hip -o tools/bf16x3_probe && tools/bf16x3_probe
// 1. numerics: Y = b + W X (W ~ N(0,1), X in (0,1): the F1 product's operands) by the exact f32 MFMA chain and by
//    three-way bf16 splits of both operands on v_mfma_f32_32x32x16_bf16 with f32 accumulation, against f64:
//    max and rms error in units of sum_k |w||x| * 2^-24, per variant (which terms, in which order).
//    It also proves the operand maps: the T-layout accumulator tile as B operand under the k permutation.
// 2. timing: the F1 segment (operand fetch, products, sigmoid of the tile, result feeds the next round) as the exact
//    kernel has it and in the bf16x3 form, at one and two waves per SIMD.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
struct Pieces { u32x4 hi[2], mid[2], lo[2]; };  // [k-step]: elements 8s..8s+7 of the 16-element operand, pairwise packed
// x = hi + mid + lo exactly (round-to-nearest pieces of 8 significant bits each)
__device__ __forceinline__ void split16(const f32x16& v, Pieces& P) {
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const float a = v[8 * s + 2 * d], b = v[8 * s + 2 * d + 1];
      const unsigned h = pk_bf16(a, b);
      const float ra = a - __builtin_bit_cast(float, h << 16), rb = b - __builtin_bit_cast(float, h & 0xffff0000u);
      const unsigned m = pk_bf16(ra, rb);
      const float sa = ra - __builtin_bit_cast(float, m << 16), sb = rb - __builtin_bit_cast(float, m & 0xffff0000u);
      P.hi[s][d] = h;
      P.mid[s][d] = m;
      P.lo[s][d] = pk_bf16(sa, sb);
    }
}
__device__ __forceinline__ f32x16 mm(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// VAR: 0 exact f32 chain; 1 six terms, small first; 2 six terms, large first; 3 nine terms, small first;
//      4 three terms (hi,hi),(hi,mid),(mid,hi) small first; 5 six terms small first, bias added last on the VALU
template <int VAR>
__device__ __forceinline__ f32x16 product(const f32x16& Wr, const f32x16& X, const f32x16& bias) {
  f32x16 acc = bias;
  if (VAR == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Wr[r], X[r], acc, 0, 0, 0);
    return acc;
  }
  Pieces A, B;
  split16(Wr, A);
  split16(X, B);
  if (VAR == 5)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    if (VAR == 1 || VAR == 5) {
      acc = mm(A.hi[s], B.lo[s], acc);
      acc = mm(A.lo[s], B.hi[s], acc);
      acc = mm(A.mid[s], B.mid[s], acc);
    } else if (VAR == 3) {
      acc = mm(A.lo[s], B.lo[s], acc);
      acc = mm(A.mid[s], B.lo[s], acc);
      acc = mm(A.lo[s], B.mid[s], acc);
      acc = mm(A.hi[s], B.lo[s], acc);
      acc = mm(A.lo[s], B.hi[s], acc);
      acc = mm(A.mid[s], B.mid[s], acc);
    }
  }
  if (VAR == 2) {
#pragma unroll
    for (int s = 0; s < 2; ++s) acc = mm(A.hi[s], B.hi[s], acc);
#pragma unroll
    for (int s = 0; s < 2; ++s) { acc = mm(A.hi[s], B.mid[s], acc); acc = mm(A.mid[s], B.hi[s], acc); }
#pragma unroll
    for (int s = 0; s < 2; ++s) { acc = mm(A.mid[s], B.mid[s], acc); acc = mm(A.hi[s], B.lo[s], acc); acc = mm(A.lo[s], B.hi[s], acc); }
    return acc;
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) { acc = mm(A.hi[s], B.mid[s], acc); acc = mm(A.mid[s], B.hi[s], acc); }
#pragma unroll
  for (int s = 0; s < 2; ++s) acc = mm(A.hi[s], B.hi[s], acc);
  if (VAR == 5)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += bias[r];
  return acc;
}

// W [nb][32][32] (out, in), X [nb][32][32] (feature k, row n), b [nb][32]; Y [nb][32][32] (out f, row n)
template <int VAR>
__global__ void k_num(const float* W, const float* X, const float* b, float* Y) {
  const int lane = threadIdx.x, c = lane & 31, h = lane >> 5;
  const float* w = W + (size_t)blockIdx.x * 1024;
  const float* x = X + (size_t)blockIdx.x * 1024;
  f32x16 Wr, Xt, bias;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int f = 8 * (r >> 2) + 4 * h + (r & 3);
    Wr[r] = w[c * 32 + f];   // A operand: row (out) c, k = f
    Xt[r] = x[f * 32 + c];   // T layout: feature f, row c
    bias[r] = b[blockIdx.x * 32 + f];
  }
  const f32x16 acc = product<VAR>(Wr, Xt, bias);
#pragma unroll
  for (int r = 0; r < 16; ++r) Y[(size_t)blockIdx.x * 1024 + (8 * (r >> 2) + 4 * h + (r & 3)) * 32 + c] = acc[r];
}

// ---- timing: R rounds of  acc = bias; acc += W H (operands of W from LDS); H = sigmoid(acc)
__device__ __forceinline__ f32x16 sigmoid_tile(const f32x16& a) {
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a[r]));
  return o;
}
template <int VAR>
__global__ void __launch_bounds__(512, 2) k_time(const float* W, float* out, int R) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, h = lane >> 5;
  float* lw = smem + wave * 2048;
  // f32 image [32][36] (exact form) / six private 16-byte slots per lane (bf16x3 form)
  f32x16 Wr;
#pragma unroll
  for (int r = 0; r < 16; ++r) Wr[r] = W[c * 32 + 8 * (r >> 2) + 4 * h + (r & 3)] * 0.3f;
  if (VAR == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) lw[c * 36 + 8 * (r >> 2) + 4 * h + (r & 3)] = Wr[r];
  } else {
    Pieces A;
    split16(Wr, A);
    u32x4* pv = reinterpret_cast<u32x4*>(lw);
    pv[0 * 64 + lane] = A.hi[0]; pv[1 * 64 + lane] = A.hi[1];
    pv[2 * 64 + lane] = A.mid[0]; pv[3 * 64 + lane] = A.mid[1];
    pv[4 * 64 + lane] = A.lo[0]; pv[5 * 64 + lane] = A.lo[1];
  }
  __syncthreads();
  f32x16 H, bias;
#pragma unroll
  for (int r = 0; r < 16; ++r) { H[r] = 0.3f + 0.01f * r + 0.001f * lane; bias[r] = 0.01f * r; }
  for (int it = 0; it < R; ++it) {
    f32x16 acc = bias;
    if (VAR == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(lw + c * 36 + 8 * q + 4 * h);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[j], H[4 * q + j], acc, 0, 0, 0);
      }
    } else {
      Pieces B;
      split16(H, B);
      const u32x4* pv = reinterpret_cast<const u32x4*>(lw);
      const u32x4 ah0 = pv[0 * 64 + lane], ah1 = pv[1 * 64 + lane], am0 = pv[2 * 64 + lane], am1 = pv[3 * 64 + lane],
                  al0 = pv[4 * 64 + lane], al1 = pv[5 * 64 + lane];
      acc = mm(ah0, B.lo[0], acc); acc = mm(al0, B.hi[0], acc); acc = mm(am0, B.mid[0], acc);
      acc = mm(ah1, B.lo[1], acc); acc = mm(al1, B.hi[1], acc); acc = mm(am1, B.mid[1], acc);
      acc = mm(ah0, B.mid[0], acc); acc = mm(am0, B.hi[0], acc);
      acc = mm(ah1, B.mid[1], acc); acc = mm(am1, B.hi[1], acc);
      acc = mm(ah0, B.hi[0], acc); acc = mm(ah1, B.hi[1], acc);
    }
    H = sigmoid_tile(acc);
  }
  float s = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += H[r];
  out[blockIdx.x * blockDim.x + tid] = s;
}

template <int VAR>
static double time_variant(const float* dW, float* dout, int threads, int R) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t lds = (size_t)(threads / 64) * 2048 * 4;
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k_time<VAR>), dim3(256), dim3(threads), lds, 0, dW, dout, R);
  CK(hipEventRecord(e0));
  for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((k_time<VAR>), dim3(256), dim3(threads), lds, 0, dW, dout, R);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / 5 * 1e6 / R;  // ns per round
}

template <int VAR>
static void run_num(const char* name, int nb, const float* dW, const float* dX, const float* db, float* dY,
                    const std::vector<double>& ref, const std::vector<double>& mag) {
  hipLaunchKernelGGL((k_num<VAR>), dim3(nb), dim3(64), 0, 0, dW, dX, db, dY);
  std::vector<float> y((size_t)nb * 1024);
  CK(hipMemcpy(y.data(), dY, y.size() * 4, hipMemcpyDeviceToHost));
  double mx = 0, sq = 0, mxrel = 0;
  for (size_t i = 0; i < y.size(); ++i) {
    const double e = std::fabs((double)y[i] - ref[i]) / (mag[i] * 5.9604644775390625e-08);
    if (e > mx) mx = e;
    sq += e * e;
    const double rl = std::fabs((double)y[i] - ref[i]) / std::max(1e-30, std::fabs(ref[i]));
    if (rl > mxrel) mxrel = rl;
  }
  printf("%-44s max %8.3f  rms %7.4f  (units of 2^-24 sum|w||x|)   max rel %.3e\n", name, mx, std::sqrt(sq / y.size()), mxrel);
}

int main() {
  const int nb = 2048;
  std::mt19937_64 g(1);
  std::normal_distribution<float> nd(0.0f, 1.0f);
  std::uniform_real_distribution<float> ud(0.0f, 1.0f);
  std::vector<float> W((size_t)nb * 1024), X((size_t)nb * 1024), b((size_t)nb * 32);
  for (auto& v : W) v = nd(g) * 1.7320508f * -1.4426950408889634f;
  for (auto& v : X) v = ud(g);
  for (auto& v : b) v = nd(g);
  // second half of the problems: delta-like operands of widely varying magnitude (the backward products)
  for (size_t i = W.size() / 2; i < W.size(); ++i) W[i] = nd(g) * std::exp(4.0f * nd(g));
  for (size_t i = X.size() / 2; i < X.size(); ++i) X[i] = nd(g) * std::exp(3.0f * nd(g)) * 1e-3f;
  for (size_t i = b.size() / 2; i < b.size(); ++i) b[i] = 0.0f;
  std::vector<double> ref((size_t)nb * 1024), mag((size_t)nb * 1024);
  for (int p = 0; p < nb; ++p)
    for (int f = 0; f < 32; ++f)
      for (int n = 0; n < 32; ++n) {
        double s = b[p * 32 + f], m = std::fabs((double)b[p * 32 + f]);
        for (int k = 0; k < 32; ++k) {
          const double t = (double)W[(size_t)p * 1024 + f * 32 + k] * (double)X[(size_t)p * 1024 + k * 32 + n];
          s += t;
          m += std::fabs(t);
        }
        ref[(size_t)p * 1024 + f * 32 + n] = s;
        mag[(size_t)p * 1024 + f * 32 + n] = m;
      }
  float *dW, *dX, *db, *dY;
  CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&db, b.size() * 4));
  CK(hipMalloc(&dY, W.size() * 4));
  CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
  printf("numerics over %d products of 32x32x32 (first half: F1-like operands, second half: wide-range operands)\n", nb);
  run_num<0>("exact: v_mfma_f32_32x32x2_f32 chain", nb, dW, dX, db, dY, ref, mag);
  run_num<1>("bf16x3, 6 terms, small first", nb, dW, dX, db, dY, ref, mag);
  run_num<2>("bf16x3, 6 terms, large first", nb, dW, dX, db, dY, ref, mag);
  run_num<3>("bf16x3, 9 terms, small first", nb, dW, dX, db, dY, ref, mag);
  run_num<4>("bf16x2-ish, 3 terms", nb, dW, dX, db, dY, ref, mag);
  run_num<5>("bf16x3, 6 terms, small first, bias last", nb, dW, dX, db, dY, ref, mag);

  float* dout;
  CK(hipMalloc(&dout, 256 * 512 * 4));
  const int R = 20000;
  for (int threads : {256, 512}) {
    const double t0 = time_variant<0>(dW, dout, threads, R);
    const double t1 = time_variant<1>(dW, dout, threads, R);
    printf("F1 segment, %d waves per SIMD: exact %.1f ns per round per wave slot, bf16x3 %.1f ns (x%.2f);"
           " per SIMD-round: %.1f vs %.1f ns\n", threads / 256, t0, t1, t0 / t1, t0 / (threads / 256), t1 / (threads / 256));
  }
  return 0;
}
