// Layerwise batched-GEMM path (f32 and f64) for models whose parameters do not fit the LDS of a CU -- BASELINE config
// 5's MLP(784-128-10) has P = 101 770 (407 KB per chain per state vector).  theta, momentum, gradient and all
// activations live in HBM; every layer of every chain is one tile job of a chain-batched GEMM on the matrix cores:
//
//   forward    H_{l+1}[c] = act(H_l[c] W_l[c]^T + b_l[c])            M = rows, N = d_{l+1}, K = d_l   (H_0 = X, shared)
//   dW         dW_l[c]    = delta_{l+1}[c]^T H_l[c]                  M = d_{l+1}, N = d_l, K = rows
//   dH         delta_l[c] = (delta_{l+1}[c] W_l[c]) * act'(H_l[c])   M = rows, N = d_l, K = d_{l+1}
//
// with the same semantics as the other kernel families (MLP.forward eeyore/models/mlp.py:45-50, losses
// eeyore/constants/constants.py:15-18, log_target eeyore/models/bayesian_model.py:30-56, gradient
// eeyore/models/log_target_model.py:15-23, HMC eeyore/samplers/hmc.py:100-156).  f32: 128x128x16 tiles, each wave a
// 64x64 quadrant of v_mfma_f32_32x32x2_f32 tiles, operands staged HBM -> LDS by the DMA path (k_bgemm_dma) or through
// registers (k_bgemm: any strides, narrow shapes); the narrow last layer fused into one pass (k_tail); the leapfrog
// update in the epilogues of the gradient kernels.  f64: k_bgemm_f64 on v_mfma_f64_16x16x4_f64.  The elementwise
// kernels and the host logic are written once for both types.
#include <atomic>
#include <type_traits>
#include <vector>

#include "ey_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ey_debug_set_variant bit 5 (A/B runs): 0 = LDS-DMA staging where the operands allow it, 1 = the register-staged kernel
#define g_bgemm_dma (!EY_VBIT(5))

#define BK 16
#define LDT(R) ((R) + 4)  // [k][row] image of an R-row operand tile: 16-byte aligned rows, staggered over banks

template <class T>
struct BGT {
  const T* A; const T* B; T* C;
  int M, N, K;
  long sAm, sAk, sBk, sBn, sCm, sCn;  // element strides
  long bA, bB, bC;                    // batch strides (0 = shared operand)
  const T* bias; long bBias;      // bias along n, per batch (nullable)
  int act;                            // activation applied to acc + bias
  const T* Hm; long sHm, sHn, bH; // if set: C = acc * act'(Hm[m][n]) with act_h
  int act_h;
  T* rowsum; long bRow;           // if set: rowsum[m] = sum_k A[m][k] (the bias gradient of a dW product)
  // if pr_theta is set the product is a weight gradient: the output becomes (acc - (theta - mu) / sigma^2) * t, i.e.
  // the prior gradient and the temperature are applied here (indexed like C; *_b like rowsum), batch stride bC / bRow
  const T *pr_theta, *pr_mu, *pr_iv, *pr_theta_b, *pr_mu_b, *pr_iv_b, *pr_temp;
  // if lf_p is set (a weight-gradient product inside an HMC trajectory) the leapfrog update is applied where the
  // gradient element g is produced:  p += wp eps g ;  theta += wt eps p  (hmc.py:105-119), the gradient itself is
  // stored only if lf_store_g, and the workgroup leaves its partial sum of (theta_new - mu)^2 / sigma^2 in
  // lf_q[b * lf_nslots + lf_slot0 + tile index] (summed in a fixed order: reproducible).  lf_p / lf_p_b are indexed
  // like C / rowsum; theta is written through pr_theta / pr_theta_b.
  T *lf_p, *lf_p_b, *lf_q;
  const T* lf_step_vec;
  T lf_step, lf_wp, lf_wt;
  int lf_slot0, lf_nslots, lf_store_g;
  // bf16x3 form: an A operand the whole batch shares (the data matrix x in the first layer's forward product) already
  // split into its three bf16 pieces, in the staging image's own order [k-chunk][piece][row][2 granules] x 16 bytes
  // (k_bf3_presplit), pre_rows rows
  const void* pre;
  int pre_rows;
  const void* preB;  // the same for a B operand the whole batch shares (x in the first layer's weight gradient), preB_rows rows
  int preB_rows;
  // every parameter has the same prior (mu0, 1 / sigma0^2) (ey_plan_set_prior detects it): the fused update then reads no prior arrays
  int pr_uniform;
  T pr_mu0, pr_iv0;
  int epi_nobatch;  // ey_debug_set_variant bit 12: the element-by-element form of the fused update (A/B runs, tests)
};
#define EY_DEV_NOBATCH(g) (((g).epi_nobatch & 1) != 0)
using BG = BGT<float>;

__device__ __forceinline__ float l_act(int code, float g) {
  switch (code) {
    case EY_ACT_SIGMOID: return 1.0f / (1.0f + __expf(-g));
    case EY_ACT_TANH: return tanhf(g);
    case EY_ACT_RELU: return g > 0.0f ? g : 0.0f;
    default: return g;
  }
}
template <class T>
__device__ __forceinline__ T l_dact(int code, T h) {
  switch (code) {
    case EY_ACT_SIGMOID: return h * (T(1) - h);
    case EY_ACT_TANH: return T(1) - h * h;
    case EY_ACT_RELU: return h > T(0) ? T(1) : T(0);
    default: return T(1);
  }
}
// the elementwise kernels below are written once for f32 and f64: f32 keeps the fast intrinsics, f64 the library
__device__ __forceinline__ float l_exp(float x) { return __expf(x); }
__device__ __forceinline__ double l_exp(double x) { return exp(x); }
__device__ __forceinline__ float l_log(float x) { return __logf(x); }
__device__ __forceinline__ double l_log(double x) { return log(x); }
__device__ __forceinline__ float l_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double l_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float l_max(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ double l_max(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ double l_act(int code, double g) {
  switch (code) {
    case EY_ACT_SIGMOID: return 1.0 / (1.0 + exp(-g));
    case EY_ACT_TANH: return tanh(g);
    case EY_ACT_RELU: return g > 0.0 ? g : 0.0;
    default: return g;
  }
}

// One operand tile [BK][RT] (RT = 128 or 32 rows) of a strided matrix, fetched into registers (so the fetch of the next
// k-tile overlaps the MFMAs of the current one) and then written to LDS as [k][row].  Groups of 4 elements along the
// contiguous stride; a 32-row tile has 128 groups (threads 128..255 idle), a 128-row tile two groups per thread.
struct Frag { float v[8]; };

// element e of the tile -> (row, k) such that consecutive threads walk the contiguous stride
template <int RT>
__device__ __forceinline__ void tile_coord(bool kfast, int e, int& row, int& kk) {
  if (kfast) { row = e >> 4; kk = e & 15; } else { row = e & (RT - 1); kk = e / RT; }
}

// Per-thread fetch state of one operand: everything that does not change from one k-tile to the next (tile
// coordinates, the 64-bit address of the first k-tile, whether the rows are inside the matrix) is worked out once; a
// k-tile then costs one pointer offset and one bound check per group.  (f32 MFMA shares the vector ALUs with this
// arithmetic: recomputing strides with 64-bit multiplies in every k-tile cost 22 % of the whole iteration.)
template <int RT>
struct Fetcher {
  static constexpr int NG = (RT * BK + 1023) / 1024;
  const float* base[NG];  // address of this thread's group in k-tile 0 (may be out of bounds: see ok/edge)
  int kk[NG];             // k offset of the group inside a k-tile
  int row[NG];            // global row of the group's first element
  bool live[NG];          // the thread has this group (a 32-row tile occupies only half of the threads)
  int lds[NG];            // where the group goes in the [k][row] LDS image
  long sRow, sK, step;    // element strides; address step per k-tile
  int rows, K;
  bool kfast;
  bool rows_inside;       // the whole tile lies inside the matrix (wave-uniform)

  __device__ __forceinline__ void init(const float* P, long sRow_, long sK_, bool kfast_, int row0, int rows_, int K_,
                                       int tid) {
    sRow = sRow_; sK = sK_; kfast = kfast_; rows = rows_; K = K_;
    step = (long)BK * sK;
    rows_inside = row0 + RT <= rows_;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const int e = (tid + 256 * i) * 4;
      live[i] = e < RT * BK;
      int r, k;
      tile_coord<RT>(kfast, e, r, k);
      row[i] = row0 + r;
      kk[i] = k;
      lds[i] = k * LDT(RT) + r;
      base[i] = P + (long)row[i] * sRow + (long)k * sK;
    }
  }

  // registers -> LDS image [k][row]: a k-fast group is 4 k's of one row (4 scalar stores LDT apart), a row-fast group 4
  // rows of one k (one 16-byte store)
  __device__ __forceinline__ void store(float* T, const Frag& f) const {
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      if (RT * BK < 1024 && !live[i]) break;
      if (kfast) {
#pragma unroll
        for (int j = 0; j < 4; ++j) T[lds[i] + j * LDT(RT)] = f.v[4 * i + j];
      } else {
        *reinterpret_cast<float4*>(T + lds[i]) = make_float4(f.v[4 * i], f.v[4 * i + 1], f.v[4 * i + 2], f.v[4 * i + 3]);
      }
    }
  }

  __device__ __forceinline__ Frag load(int kt) const {
    Frag f;
    const int k0 = kt * BK;
    if (RT * BK >= 1024 && rows_inside && k0 + BK <= K) {
      // interior k-tile of an interior block (a scalar branch): every thread loads its groups with no per-thread test
#pragma unroll
      for (int i = 0; i < NG; ++i) {
        const float* src = base[i] + (long)kt * step;
#pragma unroll
        for (int j = 0; j < 4; ++j) f.v[4 * i + j] = src[j];
      }
      return f;
    }
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      if (!live[i]) break;
      const int gr = row[i], gk = k0 + kk[i];
      const float* src = base[i] + (long)kt * step;
      const bool inside = kfast ? (gr < rows && gk + 3 < K) : (gr + 3 < rows && gk < K);
      if (inside) {  // 4 consecutive elements along the contiguous stride, all in bounds
#pragma unroll
        for (int j = 0; j < 4; ++j) f.v[4 * i + j] = src[j];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r2 = kfast ? gr : gr + j, k2 = kfast ? gk + j : gk;
          f.v[4 * i + j] = (r2 < rows && k2 < K) ? src[kfast ? (long)j * sK : (long)j * sRow] : 0.0f;
        }
      }
    }
    return f;
  }
};

// Epilogue shared by the GEMM kernels: bias + activation (forward), act'(H) (input gradient) or the prior gradient
// and the temperature (weight gradient), and the bias gradient from the A-tile row sums.
// f32 MFMA shares the vector ALUs, so every epilogue instruction is matrix time lost: the element loop is specialised
// per kind (no per-element switch), full tiles skip the bounds tests, offsets inside one batch item are 32-bit
// (checked on the host) from a per-(i, j) base plus a scalar multiple of the row stride, and sigmoid / tanh use the
// hardware exp2 and reciprocal (1 ulp each) as the fused kernels do.
__device__ __forceinline__ float l_sigmoid_fast(float g) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * g));
}
__device__ __forceinline__ float l_tanh_fast(float g) {
  return 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * g)) - 1.0f;
}
// activation / derivative over a small register array with the switch OUTSIDE the element loop
template <int n, class T>
__device__ __forceinline__ void act_vec(int code, T (&v)[n]) {
  if constexpr (sizeof(T) == 4) {
    switch (code) {
      case EY_ACT_SIGMOID:
#pragma unroll
        for (int i = 0; i < n; ++i) v[i] = l_sigmoid_fast(v[i]);
        break;
      case EY_ACT_TANH:
#pragma unroll
        for (int i = 0; i < n; ++i) v[i] = l_tanh_fast(v[i]);
        break;
      case EY_ACT_RELU:
#pragma unroll
        for (int i = 0; i < n; ++i) v[i] = fmaxf(v[i], 0.0f);
        break;
      default: break;
    }
  } else {
    if (code == EY_ACT_NONE) return;
#pragma unroll
    for (int i = 0; i < n; ++i) v[i] = l_act(code, v[i]);
  }
}
template <int n, class T>
__device__ __forceinline__ void dact_vec(int code, const T (&h)[n], T (&o)[n]) {
  switch (code) {
    case EY_ACT_SIGMOID:
#pragma unroll
      for (int i = 0; i < n; ++i) o[i] = h[i] * (T(1) - h[i]);
      break;
    case EY_ACT_TANH:
#pragma unroll
      for (int i = 0; i < n; ++i) o[i] = T(1) - h[i] * h[i];
      break;
    case EY_ACT_RELU:
#pragma unroll
      for (int i = 0; i < n; ++i) o[i] = h[i] > T(0) ? T(1) : T(0);
      break;
    default:
#pragma unroll
      for (int i = 0; i < n; ++i) o[i] = T(1);
  }
}
#define EPI_AT(base, byte_off) (*(decltype(base))((const char*)(base) + (byte_off)))
template <int TM, int TN, bool FULL, class F>
__device__ __forceinline__ void epi_loop(const BG& g, const f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn, int c,
                                         int h, F f) {
  const unsigned sCm = 4u * (unsigned)g.sCm, sCn = 4u * (unsigned)g.sCn;  // BYTE offsets: base + zext(u32) addressing
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * (32 * TN) + 32 * j + c;
    if (!FULL && n >= g.N) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int mb = m0 + wm * (32 * TM) + 32 * i + 4 * h;
      const unsigned cb = (unsigned)mb * sCm + (unsigned)n * sCn;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dm = 8 * (r >> 2) + (r & 3);  // compile-time row offset: dm * sCm is a scalar product
        if (FULL || mb + dm < g.M) f(acc[i][j][r], mb + dm, n, cb + (unsigned)dm * sCm);
      }
    }
  }
}
// Epilogues that READ per element (the prior gradient / fused leapfrog update of a weight-gradient product: theta, p, mu,
// 1 / sigma^2; the input gradient: H), batched.  Written element by element (epi_loop) the compiler waited for every
// element's loads before it stored that element and only then issued the next element's loads -- stores and loads share
// one in-order counter (vmcnt) --: 64 serial memory round trips per lane and tile, 58 us of a workgroup's 217 in config 5's
// first-layer weight gradient (the product alone 5.08 ms, with the update 6.94; tools/dw0_alone.py, tools/isa_mem_blocks.py),
// and nearly ALL of a workgroup's 41 .. 64 us in the products of mid-size models (seven k-steps per tile).  Here the lane's
// elements go in batches of BS (NL loaded values each) whose loads are all issued before the PREVIOUS batch is computed and
// stored: two batches in flight (at most 32 registers: the accumulators leave no more at three workgroups per CU), NE / BS
// + 1 round trips.  Elements beyond the matrix (tiles that are not FULL) load from offset 0 instead and are skipped by the
// consumer -- no branch depends on the lane.  `ld(v, ci, m, n)` fills the NL values of the element at byte offset ci,
// `cf(acc, v, ci, m, n)` consumes them; the elements are visited in epi_loop's order, so sums over them (the prior
// quadratic form q) keep their order of terms: bit for bit the results of the loop (ey_debug_set_variant bit 12 = the loop).
template <int TM, int TN, bool FULL, int BS, int NL, class LD, class CF>
__device__ __forceinline__ void epi_batches(const BG& g, const f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn, int c,
                                            int h, LD ld, CF cf) {
  constexpr int NE = TM * TN * 16, NB = NE / BS;
  static_assert(NE % BS == 0, "whole batches");
  const unsigned sCm = 4u * (unsigned)g.sCm, sCn = 4u * (unsigned)g.sCn;
  const int mb0 = m0 + wm * (32 * TM) + 4 * h, nb0 = n0 + wn * (32 * TN) + c;
  const unsigned base = (unsigned)mb0 * sCm + (unsigned)nb0 * sCn;
  float buf[2][BS][NL];
  // element idx = (j TM + i) 16 + r: row mb0 + 32 i + 8 (r >> 2) + (r & 3), column nb0 + 32 j
  auto dm = [](int idx) { return 32 * ((idx / 16) % TM) + 8 * ((idx % 16) >> 2) + (idx & 3); };
  auto dn = [](int idx) { return 32 * (idx / (TM * 16)); };
  auto load = [&](int e, int b) {
#pragma unroll
    for (int k = 0; k < BS; ++k) {
      const int idx = e * BS + k, m = mb0 + dm(idx), n = nb0 + dn(idx);
      const unsigned ci = base + (unsigned)dm(idx) * sCm + (unsigned)dn(idx) * sCn;
      const bool in = FULL || (m < g.M && n < g.N);
      ld(buf[b][k], in ? ci : 0u, in ? m : 0, in ? n : 0);
    }
  };
  load(0, 0);
#pragma unroll
  for (int e = 0; e < NB; ++e) {
    if (e + 1 < NB) load(e + 1, (e + 1) & 1);
    asm volatile("" ::: "memory");  // the next batch's loads stay above this batch's stores
#pragma unroll
    for (int k = 0; k < BS; ++k) {
      const int idx = e * BS + k, m = mb0 + dm(idx), n = nb0 + dn(idx);
      const unsigned ci = base + (unsigned)dm(idx) * sCm + (unsigned)dn(idx) * sCn;
      if (FULL || (m < g.M && n < g.N)) cf(acc[(idx / 16) % TM][idx / (TM * 16)][idx % 16], buf[e & 1][k], ci, m, n);
    }
    asm volatile("" ::: "memory");
  }
}
// the weight-gradient epilogue on it: minus the prior gradient, times the temperature and (LF) the fused leapfrog update.
// UNI: one (mu, 1 / sigma^2) for all parameters (ey_plan_set_prior detects it) -- no prior loads, twice the batch.
template <int TM, int TN, bool FULL, bool UNI, bool LF>
__device__ __forceinline__ float epi_dw(const BG& g, const f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn, int c,
                                        int h, long b, float tscale, float ep, float et) {
  constexpr int NL = (LF ? 2 : 1) + (UNI ? 0 : 2), BS = UNI ? (LF ? 8 : 16) : 4;
  float* __restrict__ th = const_cast<float*>(g.pr_theta) + b * g.bC;
  float* __restrict__ pp = LF ? g.lf_p + b * g.bC : nullptr;
  float* __restrict__ Cr = g.C + b * g.bC;
  const float* __restrict__ mu = g.pr_mu;
  const float* __restrict__ iv = g.pr_iv;
  const bool store_g = !LF || g.lf_store_g != 0, move = LF && g.lf_wt != 0.0f;
  const float mu0 = g.pr_mu0, iv0 = g.pr_iv0;
  float q = 0.0f;
  epi_batches<TM, TN, FULL, BS, NL>(
      g, acc, m0, n0, wm, wn, c, h,
      [&](float (&v)[NL], unsigned ci, int, int) {
        v[0] = EPI_AT(th, ci);
        if constexpr (LF) v[1] = EPI_AT(pp, ci);
        if constexpr (!UNI) { v[NL - 2] = EPI_AT(mu, ci); v[NL - 1] = EPI_AT(iv, ci); }
      },
      [&](float a, const float (&v)[NL], unsigned ci, int, int) {
        const float m_ = UNI ? mu0 : v[NL - 2], i_ = UNI ? iv0 : v[NL - 1];
        float tv = v[0];
        const float gv = (a - (tv - m_) * i_) * tscale;
        if (store_g) EPI_AT(Cr, ci) = gv;
        if constexpr (LF) {
          const float pv = v[1] + ep * gv;
          EPI_AT(pp, ci) = pv;
          if (move) { tv = tv + et * pv; EPI_AT(th, ci) = tv; }
          const float d = tv - m_;
          q += d * d * i_;
        }
      });
  return q;
}
// returns this lane's part of the prior quadratic form of the NEW position when the leapfrog update is fused in
template <int TM, int TN, bool FULL>
__device__ __forceinline__ float epi_kind(const BG& g, const f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn, int c,
                                          int h, long b, float tscale, float ep, float et) {
  float* C = g.C + b * g.bC;
  float q = 0.0f;
  if (g.pr_theta && !EY_DEV_NOBATCH(g)) {
    if (g.lf_p)
      return g.pr_uniform ? epi_dw<TM, TN, FULL, true, true>(g, acc, m0, n0, wm, wn, c, h, b, tscale, ep, et)
                          : epi_dw<TM, TN, FULL, false, true>(g, acc, m0, n0, wm, wn, c, h, b, tscale, ep, et);
    return g.pr_uniform ? epi_dw<TM, TN, FULL, true, false>(g, acc, m0, n0, wm, wn, c, h, b, tscale, ep, et)
                        : epi_dw<TM, TN, FULL, false, false>(g, acc, m0, n0, wm, wn, c, h, b, tscale, ep, et);
  } else if (g.Hm && !EY_DEV_NOBATCH(g)) {  // input gradient: times act'(H), H read in batches of 16
    float* __restrict__ Cr = C;
    const float* __restrict__ Hm = g.Hm + b * g.bH;
    const unsigned sHm = 4u * (unsigned)g.sHm, sHn = 4u * (unsigned)g.sHn;
    const int act_h = g.act_h;
    epi_batches<TM, TN, FULL, 16, 1>(
        g, acc, m0, n0, wm, wn, c, h,
        [&](float (&v)[1], unsigned, int m, int n) { v[0] = EPI_AT(Hm, (unsigned)m * sHm + (unsigned)n * sHn); },
        [&](float a, const float (&v)[1], unsigned ci, int, int) { EPI_AT(Cr, ci) = a * l_dact(act_h, v[0]); });
    return 0.0f;
  } else if (g.pr_theta && g.lf_p) {  // weight gradient with the leapfrog update fused in
    // the four arrays are distinct and every element is touched once: telling the compiler lets it issue the loads of
    // the following elements before the stores of this one (otherwise every element is a serialized HBM round trip)
    float* __restrict__ th = const_cast<float*>(g.pr_theta) + b * g.bC;
    float* __restrict__ pp = g.lf_p + b * g.bC;
    const float* __restrict__ mu = g.pr_mu;
    const float* __restrict__ iv = g.pr_iv;
    float* __restrict__ Cr = C;
    const bool store_g = g.lf_store_g != 0, move = g.lf_wt != 0.0f;
    epi_loop<TM, TN, FULL>(g, acc, m0, n0, wm, wn, c, h, [&](float v, int, int, unsigned ci) {
      const float m_ = EPI_AT(mu, ci), i_ = EPI_AT(iv, ci);
      float tv = EPI_AT(th, ci);
      const float gv = (v - (tv - m_) * i_) * tscale;
      if (store_g) EPI_AT(Cr, ci) = gv;
      const float pv = EPI_AT(pp, ci) + ep * gv;
      EPI_AT(pp, ci) = pv;
      if (move) { tv = tv + et * pv; EPI_AT(th, ci) = tv; }
      const float d = tv - m_;
      q += d * d * i_;
    });
  } else if (g.pr_theta) {  // weight gradient: minus the prior gradient, times the temperature
    const float* th = g.pr_theta + b * g.bC;
    const float* mu = g.pr_mu;
    const float* iv = g.pr_iv;
    epi_loop<TM, TN, FULL>(g, acc, m0, n0, wm, wn, c, h, [&](float v, int, int, unsigned ci) {
      EPI_AT(C, ci) = (v - (EPI_AT(th, ci) - EPI_AT(mu, ci)) * EPI_AT(iv, ci)) * tscale;
    });
  } else if (g.Hm) {  // input gradient: times act'(H)
    const float* Hm = g.Hm + b * g.bH;
    const unsigned sHm = 4u * (unsigned)g.sHm, sHn = 4u * (unsigned)g.sHn;
    const int act_h = g.act_h;
    epi_loop<TM, TN, FULL>(g, acc, m0, n0, wm, wn, c, h, [&](float v, int m, int n, unsigned ci) {
      EPI_AT(C, ci) = v * l_dact(act_h, EPI_AT(Hm, (unsigned)m * sHm + (unsigned)n * sHn));
    });
  } else {  // forward: bias, activation
    float bias[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * (32 * TN) + 32 * j + c;
      bias[j] = g.bias && n < g.N ? g.bias[b * g.bBias + n] : 0.0f;
    }
    const int nb = n0 + wn * (32 * TN) + c;
    auto bj = [&](int n) { return bias[TN == 1 ? 0 : ((n - nb) >> 5)]; };
    switch (g.act) {
      case EY_ACT_SIGMOID:
        epi_loop<TM, TN, FULL>(g, acc, m0, n0, wm, wn, c, h,
                               [&](float v, int, int n, unsigned ci) { EPI_AT(C, ci) = l_sigmoid_fast(v + bj(n)); });
        break;
      case EY_ACT_TANH:
        epi_loop<TM, TN, FULL>(g, acc, m0, n0, wm, wn, c, h,
                               [&](float v, int, int n, unsigned ci) { EPI_AT(C, ci) = l_tanh_fast(v + bj(n)); });
        break;
      case EY_ACT_RELU:
        epi_loop<TM, TN, FULL>(g, acc, m0, n0, wm, wn, c, h,
                               [&](float v, int, int n, unsigned ci) { EPI_AT(C, ci) = fmaxf(v + bj(n), 0.0f); });
        break;
      default:
        epi_loop<TM, TN, FULL>(g, acc, m0, n0, wm, wn, c, h,
                               [&](float v, int, int n, unsigned ci) { EPI_AT(C, ci) = v + bj(n); });
    }
  }
  return q;
}
template <int TM, int TN, int WGM, int WGN>
__device__ __forceinline__ void bg_epilogue(const BG& g, const f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn, int c,
                                            int h, long b, float rsum, bool do_rowsum, int tid) {
  __shared__ float epi_red[4];
  const float tscale = g.pr_temp ? g.pr_temp[b] : 1.0f;
  const bool fuse = g.lf_p != nullptr;
  float ep = 0.0f, et = 0.0f, q = 0.0f;
  if (fuse) {
    const float eps = g.lf_step_vec ? g.lf_step_vec[b] : g.lf_step;
    ep = g.lf_wp * eps;
    et = g.lf_wt * eps;
  }
  if (do_rowsum && m0 + tid < g.M) {
    const int mm = m0 + tid;
    if (fuse) {  // the bias gradient element, with the same update
      float* thb = const_cast<float*>(g.pr_theta_b) + b * g.bRow;
      const float m_ = g.pr_mu_b[mm], i_ = g.pr_iv_b[mm];
      float tv = thb[mm];
      const float gv = (rsum - (tv - m_) * i_) * tscale;
      if (g.lf_store_g) g.rowsum[b * g.bRow + mm] = gv;
      const float pv = g.lf_p_b[b * g.bRow + mm] + ep * gv;
      g.lf_p_b[b * g.bRow + mm] = pv;
      if (g.lf_wt != 0.0f) { tv = tv + et * pv; thb[mm] = tv; }
      const float d = tv - m_;
      q = d * d * i_;
    } else {
      if (g.pr_theta_b) rsum = (rsum - (g.pr_theta_b[b * g.bRow + mm] - g.pr_mu_b[mm]) * g.pr_iv_b[mm]) * tscale;
      g.rowsum[b * g.bRow + mm] = rsum;
    }
  }
  if (m0 + 32 * TM * WGM <= g.M && n0 + 32 * TN * WGN <= g.N)
    q += epi_kind<TM, TN, true>(g, acc, m0, n0, wm, wn, c, h, b, tscale, ep, et);
  else
    q += epi_kind<TM, TN, false>(g, acc, m0, n0, wm, wn, c, h, b, tscale, ep, et);
  if (fuse) {  // uniform over the workgroup
    const int gx = (g.N + 32 * TN * WGN - 1) / (32 * TN * WGN);
    const int slot = g.lf_slot0 + (m0 / (32 * TM * WGM)) * gx + n0 / (32 * TN * WGN);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    if ((tid & 63) == 0) epi_red[tid >> 6] = q;
    __syncthreads();
    if (tid == 0) g.lf_q[b * g.lf_nslots + slot] = ((epi_red[0] + epi_red[1]) + epi_red[2]) + epi_red[3];
  }
}

// Workgroups are handed to the 8 XCDs round-robin in dispatch order (x fastest), and each XCD has its own L2: the
// blocks of one batch item (one chain), which share that chain's operand, would land on 8 different L2s.  Remap the
// dispatch index so that consecutive LOGICAL blocks run on the same XCD: logical = (id % 8) * (total / 8) + id / 8.
struct BlockId { int x, y; long z; };
__device__ __forceinline__ BlockId xcd_block() {
  const unsigned gx = gridDim.x, gy = gridDim.y, gz = gridDim.z;
  const unsigned total = gx * gy * gz;
  unsigned id = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  const unsigned body = total & ~7u;
  if (id < body) id = (id & 7u) * (total >> 3) + (id >> 3);
  BlockId r;
  r.x = id % gx;
  const unsigned t = id / gx;
  r.y = t % gy;
  r.z = t / gy;
  return r;
}
// The same with the block COLUMNS divided between the two XCDs of a pair (XCD 2q takes the left half of the columns of the
// q-th quarter of the batch, XCD 2q + 1 the right half): for a product whose B operand the whole batch shares and whose pre-split
// image (4.8 MB for config 5's x) does not fit one XCD's 4 MB L2 beside the streams -- half of it does.  The per-batch operand
// is then fetched by two XCDs instead of one.  Needs an even number of block columns and a batch that divides by four; else the
// mapping above.  Config 5's first-layer weight gradient: 22.3 -> 19.9 GB of fabric traffic per dispatch, 4.51 -> 4.47 ms
// (profiles/r05_cfg5_steps.txt; that workgroup id % 8 IS the XCD: tools/xcd_probe.hip, profiles/r05_xcd_probe.txt).
__device__ __forceinline__ BlockId xcd_block_cols() {
  const unsigned gx = gridDim.x, gy = gridDim.y, gz = gridDim.z;
  const unsigned nb = gy * gz;
  if ((gx & 1u) || (nb & 3u)) return xcd_block();
  const unsigned id = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  const unsigned xcd = id & 7u, i = id >> 3, hx = gx >> 1;
  const unsigned bq = (xcd >> 1) * (nb >> 2) + i / hx;
  BlockId r;
  r.x = (xcd & 1u) * hx + i % hx;
  r.y = bq % gy;
  r.z = bq / gy;
  return r;
}

// ... and the block ROWS, for a product whose A operand the whole batch shares (x in the first layer's forward product; one block
// column): XCD 2q takes the upper half of the rows of the q-th quarter of the batch.  Config 5's forward product 3.80 -> 3.75 ms.
__device__ __forceinline__ BlockId xcd_block_rows() {
  const unsigned gx = gridDim.x, gy = gridDim.y, gz = gridDim.z;
  if (gx != 1u || (gy & 1u) || (gz & 3u)) return xcd_block();
  const unsigned id = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  const unsigned xcd = id & 7u, i = id >> 3, hy = gy >> 1;
  BlockId r;
  r.x = 0;
  r.y = (xcd & 1u) * hy + i % hy;
  r.z = (xcd >> 1) * (gz >> 2) + i / hy;
  return r;
}

// C[b] = epilogue(A[b] B[b]).  Block tile BMT x BNT x 16 with two LDS buffers; the 4 waves form a WGM x WGN grid and
// each owns TM x TN MFMA tiles of 32x32 (v_mfma_f32_32x32x2_f32), so one operand register feeds TN (or TM) MFMAs.
//   <2,2,2,2>: 128 x 128 (the big products)   <1,1,4,1>: 128 x 32 (narrow N)   <1,1,1,4>: 32 x 128 (narrow M)
template <int TM, int TN, int WGM, int WGN>
__global__ void __launch_bounds__(256) k_bgemm(BG g) {
  constexpr int BMT = 32 * TM * WGM, BNT = 32 * TN * WGN;
  __shared__ __attribute__((aligned(16))) float As[2][BK * LDT(BMT)];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDT(BNT)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 31, h = lane >> 5;
  const int wm = wave / WGN, wn = wave % WGN;
  const BlockId bid = xcd_block();
  const int m0 = bid.y * BMT, n0 = bid.x * BNT;
  const long b = bid.z;
  const float* A = g.A + b * g.bA;
  const float* B = g.B + b * g.bB;
  const bool a_kfast = g.sAk == 1, b_kfast = g.sBk == 1;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  const int ktiles = (g.K + BK - 1) / BK;
  const bool do_rowsum = g.rowsum != nullptr && bid.x == 0 && tid < BMT;
  float rsum = 0.0f;
  Fetcher<BMT> FA;
  Fetcher<BNT> FB;
  FA.init(A, g.sAm, g.sAk, a_kfast, m0, g.M, g.K, tid);
  FB.init(B, g.sBn, g.sBk, b_kfast, n0, g.N, g.K, tid);
  Frag fa = FA.load(0);
  Frag fb = FB.load(0);
  FA.store(As[0], fa);
  FB.store(Bs[0], fb);
  __syncthreads();
  for (int kt = 0; kt < ktiles; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < ktiles;
    if (more) {
      fa = FA.load(kt + 1);
      fb = FB.load(kt + 1);
    }
    const float* Ac = As[cur] + wm * (32 * TM) + c;
    const float* Bc = Bs[cur] + wn * (32 * TN) + c;
#pragma unroll
    for (int s = 0; s < BK / 2; ++s) {
      float av[TM], bv[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) av[i] = Ac[(2 * s + h) * LDT(BMT) + 32 * i];
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[j] = Bc[(2 * s + h) * LDT(BNT) + 32 * j];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    if (do_rowsum) {
#pragma unroll
      for (int k = 0; k < BK; ++k) rsum += As[cur][k * LDT(BMT) + tid];
    }
    if (more) {
      FA.store(As[cur ^ 1], fa);
      FB.store(Bs[cur ^ 1], fb);
    }
    __syncthreads();
  }
  bg_epilogue<TM, TN, WGM, WGN>(g, acc, m0, n0, wm, wn, c, h, b, rsum, do_rowsum, tid);
}

// ---- the 128 x 128 x 16 product with the operand tiles moved HBM -> LDS by the DMA path (global_load_lds_dwordx4: no
// vector registers and, above all, no vector-ALU instructions for the staging -- f32 MFMA shares the vector ALUs, and
// the register-staged version above spends a fifth of its time on fetch arithmetic and transposing LDS stores).
// Both operands must have the same contiguous direction:
//   KFAST (k contiguous: the forward products): a 16-byte piece is 4 consecutive k of one row.  The LDS image is
//     [row][4 pieces], the piece at position p of row r being k-quad p ^ ((r >> 2) & 3) (the DMA writes 64 lanes x 16 B
//     in lane order, so the swizzle is on the SOURCE address).  Lane (c, h) fetches the quads 2q + h of its row with
//     one ds_read_b128 each (conflict-free through the swizzle): MFMA step (q, j) contracts k = 8q + 4h + j.
//   !KFAST (m / n contiguous: the weight-gradient products): a piece is 4 consecutive rows at one k, the image is
//     [k][128 rows] as it streams in, a fragment is one ds_read_b32 per step (lanes <-> consecutive rows), step s
//     contracts k = 2s + h.
// Three LDS stages; a wave waits for its own pieces of tile t with a counted s_waitcnt (the next tile's stay in flight
// across the barrier), the barrier makes every wave's pieces visible, then tile t + 2 is issued into the stage that
// was read two iterations ago.  Requires K % 16 == 0; rows beyond M / N fetch the last row again (their products land
// in outputs the epilogue discards).
#define DMA_STAGES 3
template <bool KFAST>
__global__ void __launch_bounds__(256, 3) k_bgemm_dma(BG g) {
  __shared__ __attribute__((aligned(16))) float As[DMA_STAGES][BK * 128];
  __shared__ __attribute__((aligned(16))) float Bs[DMA_STAGES][BK * 128];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const BlockId bid = xcd_block();
  const int m0 = bid.y * 128, n0 = bid.x * 128;
  const long b = bid.z;
  const float* A = g.A + b * g.bA;
  const float* B = g.B + b * g.bB;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  const int ktiles = g.K / BK;
  const bool do_rowsum = g.rowsum != nullptr && bid.x == 0 && tid < 128;
  float rsum = 0.0f;

  // this lane's two pieces of every A tile and of every B tile: wave-instruction w2 = 2 wave + i covers slots 64 w2 ..
  // Rows beyond M / N are CLAMPED to the last row, not skipped: every wave then issues exactly four loads per tile (the
  // counted s_waitcnt below relies on that -- a wave whose rows are all out of range would otherwise issue fewer and
  // stop waiting for its own tile), and the loop carries no predicates.  The duplicates feed outputs the epilogue drops.
  const float* srcA[2];
  const float* srcB[2];
  long stepA, stepB;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int slot = 64 * (2 * wave + i) + lane;
    if (KFAST) {
      const int r = slot >> 2, p = slot & 3, kq = p ^ ((r >> 2) & 3);
      srcA[i] = A + (long)min(m0 + r, g.M - 1) * g.sAm + 4 * kq;
      srcB[i] = B + (long)min(n0 + r, g.N - 1) * g.sBn + 4 * kq;
    } else {
      const int k = slot >> 5, rq = slot & 31;  // M, N are multiples of 4 here (checked on the host)
      srcA[i] = A + (long)k * g.sAk + min(m0 + 4 * rq, g.M - 4);
      srcB[i] = B + (long)k * g.sBk + min(n0 + 4 * rq, g.N - 4);
    }
  }
  stepA = KFAST ? BK : (long)BK * g.sAk;
  stepB = KFAST ? BK : (long)BK * g.sBk;
  auto issue = [&](int kt, int st) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int w2 = 2 * wave + i;
      __builtin_amdgcn_global_load_lds(srcA[i] + (long)kt * stepA,
                                       (__attribute__((address_space(3))) void*)(As[st] + 256 * w2), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(srcB[i] + (long)kt * stepB,
                                       (__attribute__((address_space(3))) void*)(Bs[st] + 256 * w2), 16, 0, 0);
    }
  };
  // LDS byte addresses of this lane's fragments inside a stage (see the layouts above)
  const uint32_t lds_a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)&As[0][0];
  const uint32_t lds_b = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)&Bs[0][0];
  uint32_t offA[2], offB[2];
  if (KFAST) {
    const int ra = wm * 64 + c, rb = wn * 64 + c;  // the second 32-row tile is 2048 bytes further (same swizzle)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      offA[q] = 4 * (ra * 16 + 4 * ((2 * q + h) ^ ((ra >> 2) & 3)));
      offB[q] = 4 * (rb * 16 + 4 * ((2 * q + h) ^ ((rb >> 2) & 3)));
    }
  } else {
    offA[0] = offA[1] = 4 * (h * 128 + wm * 64 + c);
    offB[0] = offB[1] = 4 * (h * 128 + wn * 64 + c);
  }
  issue(0, 0);
  if (ktiles > 1) issue(1, 1);
  for (int kt = 0; kt < ktiles; ++kt) {
    const int st = kt % DMA_STAGES;
    // this wave's pieces of tile kt have landed (those of tile kt + 1, issued later, may still be in flight)
    if (kt + 1 < ktiles) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < ktiles) issue(kt + 2, (kt + 2) % DMA_STAGES);
    // The fragment reads are written as instructions with their own wait: the compiler cannot tell that a read of
    // stage st does not touch the stage a DMA load issued above is still filling, and would put s_waitcnt vmcnt(0)
    // in front of every compiler-visible LDS read -- serialising the fetch of tile kt + 2 with the products of tile kt.
    const uint32_t a_st = lds_a + st * (BK * 128 * 4), b_st = lds_b + st * (BK * 128 * 4);
    if (KFAST) {
      typedef float f4 __attribute__((ext_vector_type(4)));
      f4 av[2][2], bv[2][2];  // [tile][q]
      asm volatile(
          "ds_read_b128 %0, %8\n\tds_read_b128 %4, %10\n\tds_read_b128 %1, %8 offset:2048\n\t"
          "ds_read_b128 %5, %10 offset:2048\n\tds_read_b128 %2, %9\n\tds_read_b128 %6, %11\n\t"
          "ds_read_b128 %3, %9 offset:2048\n\tds_read_b128 %7, %11 offset:2048\n\ts_waitcnt lgkmcnt(0)"
          : "=&v"(av[0][0]), "=&v"(av[1][0]), "=&v"(av[0][1]), "=&v"(av[1][1]), "=&v"(bv[0][0]), "=&v"(bv[1][0]),
            "=&v"(bv[0][1]), "=&v"(bv[1][1])
          : "v"(a_st + offA[0]), "v"(a_st + offA[1]), "v"(b_st + offB[0]), "v"(b_st + offB[1]));
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][q][j4], bv[j][q][j4], acc[i][j], 0, 0, 0);
      if (do_rowsum) {  // sum over the tile's k of A[m = tid][k]: the four pieces of row tid, whatever their order
        f4 r0, r1, r2, r3;
        asm volatile(
            "ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\t"
            "ds_read_b128 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
            : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
            : "v"(a_st + tid * 64));
        // the order the compiler-visible loop had: k-positions 0..15 of the image, left to right
        rsum += r0[0]; rsum += r0[1]; rsum += r0[2]; rsum += r0[3];
        rsum += r1[0]; rsum += r1[1]; rsum += r1[2]; rsum += r1[3];
        rsum += r2[0]; rsum += r2[1]; rsum += r2[2]; rsum += r2[3];
        rsum += r3[0]; rsum += r3[1]; rsum += r3[2]; rsum += r3[3];
      }
    } else {
#pragma unroll
      for (int half = 0; half < 2; ++half) {  // 4 k-steps (16 single-dword fragments) per wait
        float av[4][2], bv[4][2];
        const uint32_t ra = a_st + offA[0] + half * 4096, rb = b_st + offB[0] + half * 4096;
        asm volatile(
            "ds_read_b32 %0, %16\n\tds_read_b32 %1, %16 offset:128\n\tds_read_b32 %8, %17\n\t"
            "ds_read_b32 %9, %17 offset:128\n\t"
            "ds_read_b32 %2, %16 offset:1024\n\tds_read_b32 %3, %16 offset:1152\n\tds_read_b32 %10, %17 offset:1024\n\t"
            "ds_read_b32 %11, %17 offset:1152\n\t"
            "ds_read_b32 %4, %16 offset:2048\n\tds_read_b32 %5, %16 offset:2176\n\tds_read_b32 %12, %17 offset:2048\n\t"
            "ds_read_b32 %13, %17 offset:2176\n\t"
            "ds_read_b32 %6, %16 offset:3072\n\tds_read_b32 %7, %16 offset:3200\n\tds_read_b32 %14, %17 offset:3072\n\t"
            "ds_read_b32 %15, %17 offset:3200\n\ts_waitcnt lgkmcnt(0)"
            : "=&v"(av[0][0]), "=&v"(av[0][1]), "=&v"(av[1][0]), "=&v"(av[1][1]), "=&v"(av[2][0]), "=&v"(av[2][1]),
              "=&v"(av[3][0]), "=&v"(av[3][1]), "=&v"(bv[0][0]), "=&v"(bv[0][1]), "=&v"(bv[1][0]), "=&v"(bv[1][1]),
              "=&v"(bv[2][0]), "=&v"(bv[2][1]), "=&v"(bv[3][0]), "=&v"(bv[3][1])
            : "v"(ra), "v"(rb));
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s2][i], bv[s2][j], acc[i][j], 0, 0, 0);
      }
      if (do_rowsum) {
        float r[BK];
        asm volatile(
            "ds_read_b32 %0, %16\n\tds_read_b32 %1, %16 offset:512\n\tds_read_b32 %2, %16 offset:1024\n\t"
            "ds_read_b32 %3, %16 offset:1536\n\tds_read_b32 %4, %16 offset:2048\n\tds_read_b32 %5, %16 offset:2560\n\t"
            "ds_read_b32 %6, %16 offset:3072\n\tds_read_b32 %7, %16 offset:3584\n\tds_read_b32 %8, %16 offset:4096\n\t"
            "ds_read_b32 %9, %16 offset:4608\n\tds_read_b32 %10, %16 offset:5120\n\tds_read_b32 %11, %16 offset:5632\n\t"
            "ds_read_b32 %12, %16 offset:6144\n\tds_read_b32 %13, %16 offset:6656\n\tds_read_b32 %14, %16 offset:7168\n\t"
            "ds_read_b32 %15, %16 offset:7680\n\ts_waitcnt lgkmcnt(0)"
            : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]),
              "=&v"(r[8]), "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11]), "=&v"(r[12]), "=&v"(r[13]), "=&v"(r[14]),
              "=&v"(r[15])
            : "v"(a_st + tid * 4));
#pragma unroll
        for (int k = 0; k < BK; ++k) rsum += r[k];
      }
    }
  }
  bg_epilogue<2, 2, 2, 2>(g, acc, m0, n0, wm, wn, c, h, b, rsum, do_rowsum, tid);
}

// ---- the 128 x 128 product in the bf16x3 form (EY_OPT_F32_PRODUCTS = EY_PRODUCTS_BF16X3, the default): every f32
// operand element is split EXACTLY into three bf16 pieces while it is staged (x = hi + mid + lo, round-to-nearest pieces
// of 8 significant bits) and a b is summed from the six piece products of relative size >= 2^-18 on
// v_mfma_f32_32x32x16_bf16, f32 accumulation, smallest terms first -- as the fused trajectory kernel does
// (ey_mfma32.hip, where the error against f64 is measured: no larger than the f32 fma chain's).  Per 16-deep k-chunk and
// wave that is 24 MFMAs of 32 cycles where the f32 form issues 32 of 64; the split costs each thread 16 elements per
// chunk (both operands), amortised over the 128-wide tile.
// Staging is through registers, either operand in either direction (AK / BK_: k contiguous, else rows contiguous): a
// thread takes one (row, 8 consecutive k) task per operand and chunk -- two 16-byte loads when k is contiguous, eight
// dword loads (lanes <-> consecutive rows) when the rows are -- and writes its three 16-byte granules into the image
// [piece][row][2 granules], granule position g ^ BF3_SWZ(row): lane (c, h) of an MFMA reads granule h of its row with
// one conflict-free ds_read_b128 per piece.  Two LDS stages of 24 KB, one barrier per chunk.
#define BK3 16
// granule position of a row in the staging image: position = granule ^ BF3_SWZ(row), BF3_SWZ = parity of the row's bits 2
// and 3.  Measured (tools/lds_b128_probe.hip, profiles/r04_lds_b128_probe.txt): a ds_write_b128 whose lanes take consecutive
// rows (32 bytes apart: the row-contiguous staging) is conflict-free when rows 4 apart alternate the position (bit 2), a
// ds_read_b128 of the fragment pattern when rows 8 apart do (bit 3), and the parity of both serves both: round 3's bit 3 alone
// left the row-contiguous stores at 16.0 cycles instead of 13.6 (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.33 on config 5's
// weight-gradient product, now 0.000; profiles/r04_cfg5_swizzle.txt).
#ifndef EY_BF3_MINB
#define EY_BF3_MINB 3  // workgroups per CU the bf16x3 product is compiled for (three: 168 registers per lane and ~280 bytes of
                       // scratch; two, without the spills, measured 2-5 % slower on config 5: 4.99 / 4.22 against 4.89 / 4.00 ms)
#endif
// EY_BF3_TIMING (diagnostic builds, tools/bf3_phase.py): s_memtime sums per phase of the chunk loop, wave 0 of every 16th workgroup
#ifndef EY_BF3_TIMING
#define EY_BF3_TIMING 0
#endif
#if EY_BF3_TIMING
__device__ unsigned long long g_bf3_phase[16];
#define BT(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); \
                   __builtin_amdgcn_sched_barrier(0); bt_acc[i] += n_ - bt_t; bt_t = n_; } while (0)
extern "C" int ey_debug_bf3_phase_read(unsigned long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_bf3_phase), 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_bf3_phase), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
#else
#define BT(i) do { } while (0)
#endif
#ifndef EY_BF3_SWZ_MASK
#define EY_BF3_SWZ_MASK 0x0C
#endif
#define BF3_SWZ(row) (__builtin_popcount((row) & EY_BF3_SWZ_MASK) & 1)
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned l_pk_bf16(float a, float b) {
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ void l_split8(const float (&v)[8], u32x4_t& hi, u32x4_t& mid, u32x4_t& lo) {
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const float a = v[2 * d], b = v[2 * d + 1];
    const unsigned hh = l_pk_bf16(a, b);
    const float ra = a - __builtin_bit_cast(float, hh << 16), rb = b - __builtin_bit_cast(float, hh & 0xffff0000u);
    const unsigned mm = l_pk_bf16(ra, rb);
    const float sa = ra - __builtin_bit_cast(float, mm << 16), sb = rb - __builtin_bit_cast(float, mm & 0xffff0000u);
    hi[d] = hh;
    mid[d] = mm;
    lo[d] = l_pk_bf16(sa, sb);
  }
}
__device__ __forceinline__ f32x16 l_mfma_bf16(const u32x4_t& a, const u32x4_t& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
// this thread's eight elements of k-chunk kt: k = 16 kt + 8 tg + j of the row `rowp` points at.  No branch at all: with
// any (even a wave-uniform one) the compiler can no longer count the younger loads that may stay in flight when it waits
// for an older chunk and waits for everything -- which serialised the fetch of chunk kt + 2 with the products of chunk kt.
// KTAIL = 0 (K a multiple of 16, decided on the host): two 16-byte loads when k is contiguous, eight dword loads
// otherwise.  KTAIL = 1: eight dword loads whose k is clamped to K - 1.  KTAIL = 2 (K a multiple of 4): a k-contiguous
// operand keeps its 16-byte loads, a group of four that lies beyond K is read from the row's last group instead.
// bf3_zero_tail drops what lies beyond K where the values are used.  Rows beyond the matrix are clamped by the caller:
// their products land in outputs the epilogue drops.
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
template <bool KF, int KTAIL>
__device__ __forceinline__ void bf3_fetch(const float* rowp, long sK, int K, int kt, int tg, float (&v)[8]) {
  const int k0 = kt * BK3 + 8 * tg;
  if constexpr (KTAIL == 2 && KF) {
    const f32x4_u lo = *reinterpret_cast<const f32x4_u*>(rowp + min(k0, K - 4));
    const f32x4_u hi = *reinterpret_cast<const f32x4_u*>(rowp + min(k0 + 4, K - 4));
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = lo[j]; v[4 + j] = hi[j]; }
  } else if constexpr (!KTAIL) {
    const float* src = rowp + (long)k0 * sK;
    if (KF) {
      const f32x4_u lo = *reinterpret_cast<const f32x4_u*>(src), hi = *reinterpret_cast<const f32x4_u*>(src + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] = lo[j]; v[4 + j] = hi[j]; }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = src[(long)j * sK];
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = rowp[(long)min(k0 + j, K - 1) * sK];
  }
}
// The same elements addressed as (wave-uniform pointer) + (32-bit lane offset), so that the loads take their base from scalar
// registers (`global_load v, voffset, s[base]`) and a chunk's address arithmetic is scalar adds: with per-lane 64-bit pointers
// every chunk cost each wave 24 vector instructions of 64-bit address arithmetic -- v_mad_u64_u32, v_mul_lo_u32, fourteen
// v_lshl_add_u64 -- in front of its sixteen loads, 1 200 cycles of a 4 400-cycle chunk (tools/bf3_phase.py).
// base: the operand of this batch item (uniform).  KTAIL as bf3_fetch (which stays for reference: the same elements).
template <bool KF, int KTAIL>
__device__ __forceinline__ void bf3_fetch_u(const float* base, unsigned row_b, const unsigned (&row_jb)[8], long sK, int K, int kt, int tg,
                                            int ktg, float (&v)[8]) {
  // row_b: this lane's row * row stride, in bytes; tg: the lane's k-half (k contiguous); ktg: 8 * the WAVE's k-half (rows contiguous).
  // Every offset is made opaque HERE, on a copy: a zero-extension hoisted out of the loop arrives as a 64-bit lane value and the
  // loads fall back to per-lane pointers.  (Opaque in place -- no v_mov -- the offsets become loop-carried through ordered
  // statements and the schedule falls apart: config 5's weight gradient 5.96 ms against 4.68.)
  const char* bc = reinterpret_cast<const char*>(base);  // (byte offsets: base + zext(offset) is the pattern the scalar-base form needs)
  if constexpr (KF) {
    const int k0 = kt * BK3 + 8 * tg;
    if constexpr (KTAIL == 1) {  // K not a multiple of 4: dword loads, k clamped
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        unsigned o = row_b + 4u * (unsigned)min(k0 + j, K - 1);
        asm volatile("" : "+v"(o));
        v[j] = *reinterpret_cast<const float*>(bc + o);
      }
    } else {
      unsigned o1 = row_b + 4u * (unsigned)(KTAIL ? min(k0, K - 4) : k0);
      unsigned o2 = row_b + 4u * (unsigned)(KTAIL ? min(k0 + 4, K - 4) : k0 + 4);
      asm volatile("" : "+v"(o1));
      asm volatile("" : "+v"(o2));
      const f32x4_u lo = *reinterpret_cast<const f32x4_u*>(bc + o1), hi = *reinterpret_cast<const f32x4_u*>(bc + o2);
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] = lo[j]; v[4 + j] = hi[j]; }
    }
  } else if constexpr (KTAIL == 0) {  // rows contiguous, the k index is the wave's: ONE scalar base per chunk and the lane's eight
                                      // offsets row_jb[j] = row_b + 4 j sK kept in registers (eight scalar bases and one lane
                                      // offset measured slower: 4.60 against 4.47 ms on config 5's weight gradient)
    const char* ubc = bc + (long)(kt * BK3 + ktg) * sK * 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      unsigned o = row_jb[j];
      asm volatile("" : "+v"(o));
      v[j] = *reinterpret_cast<const float*>(ubc + o);
    }
  } else {  // ... with a clamped k: a scalar base per load
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = min(kt * BK3 + ktg + j, K - 1);
      unsigned o = row_b;
      asm volatile("" : "+v"(o));
      v[j] = *reinterpret_cast<const float*>(bc + (long)k * sK * 4 + o);
    }
  }
}
template <int KTAIL>
__device__ __forceinline__ void bf3_zero_tail(int K, int kt, int tg, float (&v)[8]) {
  if constexpr (KTAIL) {
    const int k0 = kt * BK3 + 8 * tg;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = k0 + j < K ? v[j] : 0.0f;
  }
}
// The shared operand split once per batch: element (row r, k) = src[r sRow + k sK]; thread (k-chunk, row, granule).
__global__ void __launch_bounds__(256) k_bf3_presplit(const float* __restrict__ src, long sRow, long sK, int R, int K,
                                                      u32x4_t* __restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int ktiles = (K + BK3 - 1) / BK3;
  if (i >= (long)ktiles * R * 2) return;
  const int gi = (int)(i & 1);
  const long t = i >> 1;
  const int row = (int)(t % R), kt = (int)(t / R);
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = kt * BK3 + 8 * gi + j;
    v[j] = k < K ? src[(long)row * sRow + (long)k * sK] : 0.0f;
  }
  u32x4_t hi, mid, lo;
  l_split8(v, hi, mid, lo);
  const long unit = (long)row * 2 + gi, chunk = (long)R * 2;
  out[(long)(kt * 3 + 0) * chunk + unit] = hi;
  out[(long)(kt * 3 + 1) * chunk + unit] = mid;
  out[(long)(kt * 3 + 2) * chunk + unit] = lo;
}
// PRE: the A operand comes pre-split from g.pre -- three coalesced 16-byte loads per thread and chunk instead of the f32
// loads and the split, which every workgroup of every chain otherwise repeats on the same data: config 5's forward
// product 4.54 -> 4.24 ms.  (The same for x as the B operand of the first layer's weight gradient measured 5 % SLOWER,
// 6.97 -> 7.35 ms: 48 bytes per task instead of 32 through an L2 that product already saturates.  Not kept.)
// PREB: the same for a shared B operand (x in the first layer's weight gradient; round 5: 4.68 -> 4.48 ms once the fetches
// took their base from scalar registers, see bf3_fetch_u -- the L2 that "saturated" in round 3 was the wave waiting to issue).
// Measured on these two kernels in round 5 and not kept: the pre-split operand's fragments loaded straight from
// the image into the operand registers (no LDS: neutral / 6 % slower), three chunks of fetches in flight (neutral), issue
// priorities by wave slot (3 % slower), sched_group_barrier pipelines of the split behind the products (neutral, then spills).
template <bool AK, bool BK_, int PRE = 0, int KTAIL = 0, int PREB = 0>
__global__ void __launch_bounds__(256, EY_BF3_MINB) k_bgemm_bf3(BG g) {
  __shared__ __attribute__((aligned(16))) u32x4_t As[2][3 * 128 * 2];
  __shared__ __attribute__((aligned(16))) u32x4_t Bs[2][3 * 128 * 2];
  __shared__ float rs_red[2][128];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const BlockId bid = PREB ? xcd_block_cols() : (PRE ? xcd_block_rows() : xcd_block());
  const int m0 = bid.y * 128, n0 = bid.x * 128;
  const long b = bid.z;
  const float* A = g.A + b * g.bA;
  const float* B = g.B + b * g.bB;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  const int ktiles = (g.K + BK3 - 1) / BK3;
  const bool want_rowsum = PRE == 0 && g.rowsum != nullptr && bid.x == 0;  // (a pre-split A is never a delta: no row sums)
  float rsum = 0.0f;
  // staging tasks: k contiguous -> consecutive lanes take the two granules of a row; rows contiguous -> consecutive lanes
  // take consecutive rows
  const int ar = AK ? (tid >> 1) : (tid & 127), ag = AK ? (tid & 1) : (tid >> 7);
  const int br = BK_ ? (tid >> 1) : (tid & 127), bg = BK_ ? (tid & 1) : (tid >> 7);
  const int a_slot = ar * 2 + (ag ^ BF3_SWZ(ar)), b_slot = br * 2 + (bg ^ BF3_SWZ(br));
  // fragment slots of this lane: rows wm * 64 + 32 i + c (A) and wn * 64 + 32 j + c (B); 32 rows further = 64 slots
  const int ra = wm * 64 + c, rb = wn * 64 + c;
  const int fa[2] = {ra * 2 + (h ^ BF3_SWZ(ra)), (ra + 32) * 2 + (h ^ BF3_SWZ(ra + 32))};
  const int fb[2] = {rb * 2 + (h ^ BF3_SWZ(rb)), (rb + 32) * 2 + (h ^ BF3_SWZ(rb + 32))};
  // (uniform base) + (lane offset) addressing of the K-whole instantiations, see bf3_fetch_u
  const unsigned a_rowb = 4u * (unsigned)(min(m0 + ar, g.M - 1) * g.sAm), b_rowb = 4u * (unsigned)(min(n0 + br, g.N - 1) * g.sBn);
  unsigned a_rowjb[8], b_rowjb[8];  // (only the rows-contiguous K-whole instantiations keep them)
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    a_rowjb[j] = a_rowb + 4u * (unsigned)(j * g.sAk);
    b_rowjb[j] = b_rowb + 4u * (unsigned)(j * g.sBk);
  }
  const int a_ktg = AK ? 0 : 8 * __builtin_amdgcn_readfirstlane(ag);
  const int b_ktg = BK_ ? 0 : 8 * __builtin_amdgcn_readfirstlane(bg);
  // Two chunks of fetches are in flight: chunk kt + 2 is requested at the top of iteration kt, chunk kt + 1 (requested an
  // iteration earlier) is split and staged at its end.  With one chunk ahead every iteration waited for the fetch it had just
  // issued (s_waitcnt vmcnt(0) behind 24 MFMAs = 0.3 us of cover for a 1 - 2 us round trip): the matrix pipe was busy 36 - 47 %
  // of config 5's two big products.  The chunks alternate between two register sets (the loop is unrolled by two).
  float va[2][8], vb[2][8];
  // the pre-split operand: thread (row tid >> 1, granule tid & 1), consecutive threads consecutive 16-byte units
  const int pr = tid >> 1, pg = tid & 1;
  const int p_slot = pr * 2 + (pg ^ BF3_SWZ(pr));
  const long pre_chunk = (long)g.pre_rows * 2;  // 16-byte units per (k-chunk, piece)
  const u32x4_t* pre_img = reinterpret_cast<const u32x4_t*>(g.pre);
  const unsigned pre_off = PRE != 1 ? 0u : 16u * (unsigned)(min(m0 + pr, g.M - 1) * 2 + pg);
  u32x4_t vp[2][3];
  const long preb_chunk = (long)g.preB_rows * 2;
  const u32x4_t* preb_img = reinterpret_cast<const u32x4_t*>(g.preB);
  const unsigned preb_off = PREB != 1 ? 0u : 16u * (unsigned)(min(n0 + pr, g.N - 1) * 2 + pg);
  u32x4_t vpb[2][3];
  auto fetch = [&](int kt, auto set_tag) {
    constexpr int S = decltype(set_tag)::value;
    if constexpr (PRE == 1) {
      unsigned o = pre_off;  // (scalar base + lane offset, as bf3_fetch_u)
      asm volatile("" : "+v"(o));
#pragma unroll
      for (int p = 0; p < 3; ++p)
        vp[S][p] = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const char*>(pre_img + (long)(kt * 3 + p) * pre_chunk) + o);
    } else if constexpr (PRE == 0) {
      bf3_fetch_u<AK, KTAIL>(A, a_rowb, a_rowjb, g.sAk, g.K, kt, ag, a_ktg, va[S]);
    }
    if constexpr (PREB == 1) {
      unsigned o = preb_off;
      asm volatile("" : "+v"(o));
#pragma unroll
      for (int p = 0; p < 3; ++p)
        vpb[S][p] = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const char*>(preb_img + (long)(kt * 3 + p) * preb_chunk) + o);
    } else if constexpr (PREB == 0) {
      bf3_fetch_u<BK_, KTAIL>(B, b_rowb, b_rowjb, g.sBk, g.K, kt, bg, b_ktg, vb[S]);
    }
  };
  auto stage = [&](int st, int kt, auto set_tag) {  // chunk kt from register set S into LDS stage st
    constexpr int S = decltype(set_tag)::value;
    u32x4_t hi, mid, lo;
    if constexpr (!PRE) bf3_zero_tail<KTAIL>(g.K, kt, ag, va[S]);
    if constexpr (!PREB) bf3_zero_tail<KTAIL>(g.K, kt, bg, vb[S]);
    if constexpr (PRE == 1) {
      As[st][0 * 256 + p_slot] = vp[S][0]; As[st][1 * 256 + p_slot] = vp[S][1]; As[st][2 * 256 + p_slot] = vp[S][2];
    } else if constexpr (PRE == 0) {
      // (the iteration behind the last chunk stages that chunk once more: not summed twice.  A select, not a branch: a branch
      // here ends the basic block of the chunk's products, and the split below can then never be scheduled among them)
      const float rs8 = ((va[S][0] + va[S][1]) + (va[S][2] + va[S][3])) + ((va[S][4] + va[S][5]) + (va[S][6] + va[S][7]));
      rsum = __builtin_fmaf(rs8, (want_rowsum && kt < ktiles) ? 1.0f : 0.0f, rsum);  // (x 1 and + 0 are exact: the same sums)
      l_split8(va[S], hi, mid, lo);
      As[st][0 * 256 + a_slot] = hi; As[st][1 * 256 + a_slot] = mid; As[st][2 * 256 + a_slot] = lo;
    }
    if constexpr (PREB == 1) {
      Bs[st][0 * 256 + p_slot] = vpb[S][0]; Bs[st][1 * 256 + p_slot] = vpb[S][1]; Bs[st][2 * 256 + p_slot] = vpb[S][2];
    } else if constexpr (PREB == 0) {
      l_split8(vb[S], hi, mid, lo);
      Bs[st][0 * 256 + b_slot] = hi; Bs[st][1 * 256 + b_slot] = mid; Bs[st][2 * 256 + b_slot] = lo;
    }
  };
  typedef std::integral_constant<int, 0> Set0;
  typedef std::integral_constant<int, 1> Set1;
  // one iteration: chunk kt is multiplied out of LDS stage kt & 1; chunk kt + 2 goes into the register set chunk kt had,
  // chunk kt + 1 leaves the other one for LDS stage (kt + 1) & 1
#if EY_BF3_TIMING
  unsigned long long bt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long bt_t = __builtin_amdgcn_s_memtime();
  const unsigned long long bt_begin = bt_t;
#endif
  auto iteration = [&](int kt, auto mine, auto other) {
    const int cur = kt & 1;
    BT(0);  // (what lies between the barrier and here: loop control)
    // (unconditional: behind a branch the compiler can no longer count how many younger loads may stay in flight when it
    // waits for chunk kt + 1 and waits for all of them; the last two iterations fetch the last chunk again, unused)
    fetch(min(kt + 2, ktiles - 1), mine);
    BT(1);  // the fetches issued
    u32x4_t pa[2][3], pb[2][3];  // [tile][piece: hi, mid, lo]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        pa[i][p] = As[cur][p * 256 + fa[i]];
        pb[i][p] = Bs[cur][p * 256 + fb[i]];
      }
#if EY_BF3_TIMING
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int pp = 0; pp < 3; ++pp) asm volatile("" : "+v"(pa[i][pp]), "+v"(pb[i][pp]));
#endif
    BT(2);  // the fragments read
    // (hi, lo), (lo, hi), (mid, mid), (hi, mid), (mid, hi), (hi, hi): smallest terms first
    constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = l_mfma_bf16(pa[i][TA[t]], pb[j][TB[t]], acc[i][j]);
    BT(3);  // the products issued
    // (unconditional as well: the last iteration stages the re-fetched last chunk into the stage nobody reads any more;
    // behind a branch the loop header waited for every load, see fetch)
    stage(cur ^ 1, kt + 1, other);
    BT(4);  // chunk kt + 1 split and staged
    __syncthreads();
    BT(5);  // the barrier
  };
  fetch(0, Set0());
  stage(0, 0, Set0());
  __syncthreads();
  fetch(min(1, ktiles - 1), Set1());
  int kt = 0;
  for (; kt + 1 < ktiles; kt += 2) {
    iteration(kt, Set0(), Set1());
    iteration(kt + 1, Set1(), Set0());
  }
  if (kt < ktiles) iteration(kt, Set0(), Set1());
#if EY_BF3_TIMING
  const unsigned long long bt_loop_end = __builtin_amdgcn_s_memtime();
#endif
  bool do_rowsum = false;
  if (want_rowsum) {  // the two granule columns of a row, in a fixed order
    rs_red[ag][ar] = rsum;
    __syncthreads();
    do_rowsum = tid < 128;
    if (do_rowsum) rsum = rs_red[0][tid] + rs_red[1][tid];
  }
  bg_epilogue<2, 2, 2, 2>(g, acc, m0, n0, wm, wn, c, h, b, rsum, do_rowsum, tid);
#if EY_BF3_TIMING
  if (tid == 0 && (blockIdx.x + blockIdx.y * gridDim.x + blockIdx.z * gridDim.x * gridDim.y) % 16 == 0) {
    const unsigned long long bt_end = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 6; ++i) atomicAdd(&g_bf3_phase[i], bt_acc[i]);
    atomicAdd(&g_bf3_phase[8], bt_loop_end - bt_begin);  // the chunk loop
    atomicAdd(&g_bf3_phase[9], bt_end - bt_loop_end);    // row sums + epilogue
    atomicAdd(&g_bf3_phase[10], 1ull);
    atomicAdd(&g_bf3_phase[11], (unsigned long long)ktiles);
  }
#endif
}

static bool dma_ok(const BG& g, bool& kfast) {
  if (g.K % BK != 0 || g.M <= 32 || g.N <= 32) return false;
  const bool a_k = g.sAk == 1, b_k = g.sBk == 1, a_m = g.sAm == 1, b_n = g.sBn == 1;
  // The 16-byte pieces only need dword alignment (a chain's theta starts at a multiple of P floats: 8-byte aligned for
  // P = 101 770), as any global load on this device; what must hold is that a piece never straddles a row.
  if (a_k && b_k) {
    kfast = true;
    return true;
  }
  if (a_m && b_n) {
    kfast = false;
    return g.M % 4 == 0 && g.N % 4 == 0;
  }
  return false;
}

// Products with a short contraction, K <= 16, through the vector ALUs: a handful of multiply-adds per output do not
// need the matrix cores or a 128 x 128 tile -- the product is HBM-bound on its output (and on H_l for the input
// gradient).  Two users: the input gradient of a narrow layer, delta_l = (delta_{l+1} W_l) * act'(H_l) with K = d_{l+1}
// (the 10-class output layer of config 5), and the forward product of a layer with few inputs, H_1 = act(X W_0^T + b_0)
// with K = d_0 (tabular data: Iris has 4 features).  One thread per four consecutive outputs of a row; the chain's
// K x N operand is staged in LDS (whatever its strides), the row's K values of A come through the scalar path.
#define DH_ROWS 128  // rows per workgroup (the staged operand is reused across them)
template <class T>
struct alignas(16) Vec4 { T x, y, z, w; };
template <class T>
struct alignas(8) Vec2 { T x, y; };
template <class T, bool FWD>
__global__ void __launch_bounds__(256) k_smallk(BGT<T> g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wsm_raw[];
  T* wsm = reinterpret_cast<T*>(wsm_raw);  // [K][N] (+ [N] bias)
  const long b = blockIdx.z;
  const T* A = g.A + b * g.bA;
  const T* B = g.B + b * g.bB;
  T* C = g.C + b * g.bC;
  const int N4 = g.N >> 2;
  for (int i = threadIdx.x; i < g.K * g.N; i += 256) {
    const int k = i / g.N, n = i - k * g.N;
    wsm[i] = B[(long)k * g.sBk + (long)n * g.sBn];
  }
  T* bsm = wsm + g.K * g.N;
  T* asm_ = bsm + g.N;  // the workgroup's DH_ROWS rows of A, [row][K]: the inner loop then waits for LDS, not for memory
  if (FWD)
    for (int i = threadIdx.x; i < g.N; i += 256) bsm[i] = g.bias ? g.bias[b * g.bBias + i] : T(0);
  const int m_lo = blockIdx.x * DH_ROWS, m_end = min(g.M, m_lo + DH_ROWS);
  for (int i = threadIdx.x; i < (m_end - m_lo) * g.K; i += 256) {
    const int r = i / g.K, k = i - r * g.K;
    asm_[i] = A[(long)(m_lo + r) * g.sAm + k];
  }
  __syncthreads();
  const int rows_per_pass = 256 / N4;
  const int n = (threadIdx.x % N4) * 4;
  if (threadIdx.x >= rows_per_pass * N4) return;
  for (int m = m_lo + threadIdx.x / N4; m < m_end; m += rows_per_pass) {
    T acc[4] = {T(0), T(0), T(0), T(0)};
    const T* ar = asm_ + (m - m_lo) * g.K;
    for (int k = 0; k < g.K; ++k) {
      const T a = ar[k];
      const Vec4<T> w = *reinterpret_cast<const Vec4<T>*>(wsm + k * g.N + n);
      acc[0] += a * w.x; acc[1] += a * w.y; acc[2] += a * w.z; acc[3] += a * w.w;
    }
    if (FWD) {
      const Vec4<T> bv = *reinterpret_cast<const Vec4<T>*>(bsm + n);
      acc[0] += bv.x; acc[1] += bv.y; acc[2] += bv.z; acc[3] += bv.w;
      act_vec<4>(g.act, acc);
    } else {
      const Vec4<T> hv = *reinterpret_cast<const Vec4<T>*>(g.Hm + b * g.bH + (long)m * g.sHm + n);
      const T hh[4] = {hv.x, hv.y, hv.z, hv.w};
      T da[4];
      dact_vec<4>(g.act_h, hh, da);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] *= da[j];
    }
    *reinterpret_cast<Vec4<T>*>(C + (long)m * g.sCm + n) = Vec4<T>{acc[0], acc[1], acc[2], acc[3]};
  }
}
// 0 = not this kernel's product, 1 = input gradient, 2 = forward
template <class T>
static int smallk_kind(const BGT<T>& g) {
  if (g.pr_theta || g.rowsum || g.K > 16 || g.N % 4 != 0 || g.N < 4 || g.N > 1024 || g.sAk != 1 || g.sCn != 1 ||
      g.sCm % 4 != 0 || g.bC % 4 != 0 || ((uintptr_t)g.C & 15) != 0 || ((size_t)(g.K + 1) * g.N + (size_t)DH_ROWS * g.K) * sizeof(T) > 48 * 1024)
    return 0;
  if (g.Hm) {
    if (g.bias || g.sHn != 1 || g.sHm % 4 != 0 || g.bH % 4 != 0 || ((uintptr_t)g.Hm & 15) != 0) return 0;
    return 1;
  }
  return 2;
}

// Weight gradient of a layer with few inputs, dW[m][n] = sum_k delta[k][m] X[k][n] with N = d_l <= 16 (the first layer on
// tabular data) and M = d_{l+1} <= 256: HBM-bound on reading delta once.  One workgroup per chain; thread (grp, m) owns
// output row m for the rows k = grp, grp + G, ... (G = 256 / Mp groups, Mp = M rounded up to whole waves, so that the
// lanes of a wave share k and the X row is an LDS broadcast); X is staged in LDS 64 rows at a time.  The groups'
// partial sums are combined in group order (reproducible), then the epilogue of the weight-gradient products: bias
// gradient (the sum of delta over rows), prior gradient, temperature and, when asked, the leapfrog update with the
// workgroup's partial prior sum in ONE slot.
#define DWN_XROWS 64
#ifndef DWN_FLIGHT
#define DWN_FLIGHT 8  // rows of delta in flight per thread (16, and a 128-row x tile: 0.652 against 0.605 ms on config 5's sixteen remainder columns)
#endif
template <class T>
__global__ void __launch_bounds__(256) k_dw_smalln(BGT<T> g) {
  __shared__ __attribute__((aligned(16))) T xs[DWN_XROWS * 16];
  __shared__ T part[256 * 17];  // [grp][m][N + 1], reused for the reduction of q
  const int tid = threadIdx.x;
  const long b = blockIdx.x;
  const int M = g.M, N = g.N, K = g.K;
  const int Mp = (M + 63) & ~63, G = 256 / Mp;
  const int m = tid % Mp, grp = tid / Mp;
  const bool live = m < M && grp < G;
  const T* A = g.A + b * g.bA;   // delta: [k][m], m contiguous
  const T* B = g.B + b * g.bB;   // X: [k][n], n contiguous
  T acc[16], rs = T(0.0);
#pragma unroll
  for (int n = 0; n < 16; ++n) acc[n] = T(0.0);
  for (int k0 = 0; k0 < K; k0 += DWN_XROWS) {
    const int kn = min(DWN_XROWS, K - k0);
    __syncthreads();
    for (int i = tid; i < kn * 16; i += 256) {
      const int kk = i >> 4, n = i & 15;
      xs[i] = n < N ? B[(long)(k0 + kk) * g.sBk + n] : T(0.0);
    }
    __syncthreads();
    if (live) {
      // eight rows' delta values are fetched before any is used: the loop is bound by the latency of these loads
      for (int kk = grp; kk < kn; kk += DWN_FLIGHT * G) {
        T a[DWN_FLIGHT];
#pragma unroll
        for (int u = 0; u < DWN_FLIGHT; ++u) a[u] = kk + u * G < kn ? A[(long)(k0 + kk + u * G) * g.sAk + m] : T(0.0);
#pragma unroll
        for (int u = 0; u < DWN_FLIGHT; ++u) {
          if (kk + u * G >= kn) break;
          rs += a[u];
          const Vec4<T>* xr = reinterpret_cast<const Vec4<T>*>(xs + (kk + u * G) * 16);
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const Vec4<T> xv = xr[v];
            acc[4 * v] += a[u] * xv.x; acc[4 * v + 1] += a[u] * xv.y; acc[4 * v + 2] += a[u] * xv.z;
            acc[4 * v + 3] += a[u] * xv.w;
          }
        }
      }
    }
  }
  __syncthreads();
  if (live) {
#pragma unroll
    for (int n = 0; n < 16; ++n) part[(grp * Mp + m) * 17 + n] = acc[n];
    part[(grp * Mp + m) * 17 + 16] = rs;
  }
  __syncthreads();
  // ---- epilogue: thread e handles output element e of the M x N block, then the M bias elements
  const T tscale = g.pr_temp ? g.pr_temp[b] : T(1.0);
  const bool fuse = g.lf_p != nullptr;
  const T eps = fuse ? (g.lf_step_vec ? g.lf_step_vec[b] : g.lf_step) : T(0.0);
  const T ep = g.lf_wp * eps, et = g.lf_wt * eps;
  T q = T(0.0);
  auto emit = [&](T v, T* gout, T* theta, T* pmom, const T* mu, const T* iv) {
    const T m_ = *mu, i_ = *iv;
    T tv = *theta;
    const T gv = (v - (tv - m_) * i_) * tscale;
    if (!fuse) { *gout = gv; return; }
    if (g.lf_store_g) *gout = gv;
    const T pv = *pmom + ep * gv;
    *pmom = pv;
    if (g.lf_wt != T(0.0)) { tv = tv + et * pv; *theta = tv; }
    const T dd = tv - m_;
    q += dd * dd * i_;
  };
  for (int e = tid; e < M * N; e += 256) {
    const int mm = e / N, n = e - mm * N;
    T v = T(0.0);
    for (int gg = 0; gg < G; ++gg) v += part[(gg * Mp + mm) * 17 + n];
    const long ci = (long)mm * g.sCm + n;
    emit(v, g.C + b * g.bC + ci, const_cast<T*>(g.pr_theta) + b * g.bC + ci, fuse ? g.lf_p + b * g.bC + ci : nullptr,
         g.pr_mu + ci, g.pr_iv + ci);
  }
  if (g.rowsum) {
    for (int mm = tid; mm < M; mm += 256) {
      T v = T(0.0);
      for (int gg = 0; gg < G; ++gg) v += part[(gg * Mp + mm) * 17 + 16];
      if (g.pr_theta_b)
        emit(v, g.rowsum + b * g.bRow + mm, const_cast<T*>(g.pr_theta_b) + b * g.bRow + mm,
             fuse ? g.lf_p_b + b * g.bRow + mm : nullptr, g.pr_mu_b + mm, g.pr_iv_b + mm);
      else
        g.rowsum[b * g.bRow + mm] = v;
    }
  }
  if (fuse) {  // uniform over the workgroup
    __syncthreads();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    if ((tid & 63) == 0) part[tid >> 6] = q;
    __syncthreads();
    if (tid == 0) g.lf_q[b * g.lf_nslots + g.lf_slot0] = ((part[0] + part[1]) + part[2]) + part[3];
  }
}
// a weight-gradient product (pr_theta set) of a layer with at most 16 inputs and 256 outputs, both operands row-major
template <class T>
static bool dw_smalln_ok(const BGT<T>& g) {
  return g.pr_theta && !g.Hm && !g.bias && g.N <= 16 && g.M <= 256 && g.sAm == 1 && g.sBn == 1 && g.sCn == 1 &&
         (!g.rowsum || g.pr_theta_b || !g.lf_p);
}

static int bgemm_one(BG g, int batch, hipStream_t s, int* cursor, bool dry);

// One product, split where that saves padded work: a weight gradient whose N is a few columns past a multiple of 128
// (784 = 6 x 128 + 16 in config 5: the seventh 128-wide block would be 12 % full) runs its remainder through the
// 32-wide kernel.
// cursor: next free slot of the fused leapfrog update's partial sums (g.lf_p set), advanced by the blocks launched;
// dry: only advance the cursor (the host sizes the slot buffer with the same dispatch logic it launches with).
// the products that do not go through a matrix-core kernel (both dtypes); *handled says whether this was one
template <class T>
static int bgemm_narrow(const BGT<T>& g, int batch, hipStream_t s, int* cursor, bool dry, bool* handled) {
  *handled = true;
  if (dw_smalln_ok(g)) {
    BGT<T> h = g;
    if (cursor && (g.lf_p || dry)) {
      h.lf_slot0 = *cursor;
      *cursor += 1;
      if (dry) return EY_OK;
    }
    hipLaunchKernelGGL((k_dw_smalln<T>), dim3(batch), dim3(256), 0, s, h);
    EY_HIP(hipGetLastError());
    return EY_OK;
  }
  if (const int kind = smallk_kind(g)) {
    const dim3 grid((g.M + DH_ROWS - 1) / DH_ROWS, 1, batch);
    const size_t lds = ((size_t)(g.K + 1) * g.N + (size_t)DH_ROWS * g.K) * sizeof(T);
    if (kind == 1) hipLaunchKernelGGL((k_smallk<T, false>), grid, dim3(256), lds, s, g);
    else hipLaunchKernelGGL((k_smallk<T, true>), grid, dim3(256), lds, s, g);
    EY_HIP(hipGetLastError());
    return EY_OK;
  }
  *handled = false;
  return EY_OK;
}
static int bgemm(const BG& g, int batch, hipStream_t s, int* cursor = nullptr, bool dry = false) {
  {
    bool handled;
    const int rc = bgemm_narrow(g, batch, s, cursor, dry, &handled);
    if (handled) return rc;
  }
  const int rem = g.N % 128;
  if (g.M > 32 && g.N > 128 && rem > 0 && rem <= 32 && !g.bias && !g.Hm) {
    BG body = g, tail = g;
    body.N = g.N - rem;
    int rc = bgemm_one(body, batch, s, cursor, dry);
    if (rc) return rc;
    const long off = (long)body.N * g.sCn;  // the tail's columns: B, C and everything indexed like C move along n
    tail.N = rem;
    tail.B = g.B + (long)body.N * g.sBn;
    tail.preB = nullptr;
    tail.C = g.C + off;
    tail.rowsum = nullptr;  // the body's first block column has written the row sums
    if (g.pr_theta) { tail.pr_theta = g.pr_theta + off; tail.pr_mu = g.pr_mu + off; tail.pr_iv = g.pr_iv + off; }
    if (g.lf_p) { tail.lf_p = g.lf_p + off; tail.lf_p_b = nullptr; }
    {  // sixteen columns or fewer: the vector-ALU kernel that reads delta once per chain (k_dw_smalln) rather than a 32-wide f32 tile
      bool handled;
      rc = bgemm_narrow(tail, batch, s, cursor, dry, &handled);
      if (handled) return rc;
    }
    return bgemm_one(tail, batch, s, cursor, dry);
  }
  return bgemm_one(g, batch, s, cursor, dry);
}

static int bgemm_one(BG g, int batch, hipStream_t s, int* cursor, bool dry) {
  if (cursor && (g.lf_p || dry)) {
    const bool narrow_n = g.N <= 32, narrow_m = !narrow_n && g.M <= 32;
    const int gx = (g.N + (narrow_n ? 31 : 127)) / (narrow_n ? 32 : 128);
    const int gy = (g.M + (narrow_m ? 31 : 127)) / (narrow_m ? 32 : 128);
    g.lf_slot0 = *cursor;
    *cursor += gx * gy;
    if (dry) return EY_OK;
  }
  // the epilogue indexes one batch item's output (and H) with 32-bit byte offsets
  if ((g.M - 1) * g.sCm + (g.N - 1) * g.sCn >= (1L << 30) ||
      (g.Hm && (g.M - 1) * g.sHm + (g.N - 1) * g.sHn >= (1L << 30)))
    return EY_ERR_UNSUPPORTED;
  // pick the tile shape by the narrow dimension: a 32-wide tile wastes 4x less on N (or M) <= 32
  if (g.N <= 32) {
    dim3 grid((g.N + 31) / 32, (g.M + 127) / 128, batch);
    hipLaunchKernelGGL((k_bgemm<1, 1, 4, 1>), grid, dim3(256), 0, s, g);
  } else if (g.M <= 32) {
    dim3 grid((g.N + 127) / 128, (g.M + 31) / 32, batch);
    hipLaunchKernelGGL((k_bgemm<1, 1, 1, 4>), grid, dim3(256), 0, s, g);
  } else {
    dim3 grid((g.N + 127) / 128, (g.M + 127) / 128, batch);
    bool kfast = false;
    const bool a_k = g.sAk == 1, b_k = g.sBk == 1;
    // (the bf16x3 kernel addresses an operand of one batch item with 32-bit byte offsets from a scalar base, bf3_fetch_u: an
    // operand beyond 4 GB per item -- a million rows of a thousand features -- takes the f32 kernels and their 64-bit pointers)
    const bool fits32 = (g.M - 1) * g.sAm + (g.K + BK3) * g.sAk < (1L << 30) && (g.N - 1) * g.sBn + (g.K + BK3) * g.sBk < (1L << 30);
    if (t_ey_products == EY_PRODUCTS_BF16X3 && (a_k || g.sAm == 1) && (b_k || g.sBn == 1) && fits32) {
      // the tail of K (a pre-split operand is zero-padded to whole chunks: its own fetch has none)
      const int ktail = g.K % BK3 == 0 ? 0 : ((a_k || b_k) && g.K % 4 == 0 && g.K >= 4 ? 2 : 1);
#define EY_BF3_LAUNCH(A_, B_, P_)                                                                      \
  do {                                                                                                 \
    if (ktail == 2) hipLaunchKernelGGL((k_bgemm_bf3<A_, B_, P_, 2>), grid, dim3(256), 0, s, g);        \
    else if (ktail == 1) hipLaunchKernelGGL((k_bgemm_bf3<A_, B_, P_, 1>), grid, dim3(256), 0, s, g);   \
    else hipLaunchKernelGGL((k_bgemm_bf3<A_, B_, P_, 0>), grid, dim3(256), 0, s, g);                   \
  } while (0)
      if (g.preB && !a_k && !b_k) {  // (rows contiguous on both sides: K's tail is 0 or 1)
        if (ktail == 1) hipLaunchKernelGGL((k_bgemm_bf3<false, false, 0, 1, 1>), grid, dim3(256), 0, s, g);
        else hipLaunchKernelGGL((k_bgemm_bf3<false, false, 0, 0, 1>), grid, dim3(256), 0, s, g);
      } else if (g.pre && a_k && b_k && !g.rowsum) EY_BF3_LAUNCH(true, true, 1);
      else if (a_k && b_k) EY_BF3_LAUNCH(true, true, 0);
      else if (a_k) EY_BF3_LAUNCH(true, false, 0);
      else if (b_k) EY_BF3_LAUNCH(false, true, 0);
      else EY_BF3_LAUNCH(false, false, 0);
#undef EY_BF3_LAUNCH
    } else if (g_bgemm_dma && dma_ok(g, kfast)) {
      if (kfast) hipLaunchKernelGGL(k_bgemm_dma<true>, grid, dim3(256), 0, s, g);
      else hipLaunchKernelGGL(k_bgemm_dma<false>, grid, dim3(256), 0, s, g);
    } else {
      hipLaunchKernelGGL((k_bgemm<2, 2, 2, 2>), grid, dim3(256), 0, s, g);
    }
  }
  EY_HIP(hipGetLastError());
  return EY_OK;
}


// ---- f64 (the reference's default dtype, eeyore/models/model.py:7): the same three products on
// v_mfma_f64_16x16x4_f64.  64 x 64 x 8 block tile through two LDS buffers (register-staged fetch of the next k-tile
// while the current one is multiplied), the 4 waves in a 2 x 2 grid, each owning 2 x 2 MFMA tiles of 16 x 16.
// Operand lane (i = lane & 15, k = lane >> 4); accumulator register r of lane (g = lane >> 4, c = lane & 15) is element
// (row 4 r + g, column c) (profiles/r02_mfma_f64_lane_map.txt).  One generic kernel serves every shape of the path --
// the f64 path exists for parity runs of models beyond LDS, not for the headline rate.
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define BK64 8
#define LD64 66  // [k][64 rows] image, rows padded: 16-byte aligned pairs, k-rows staggered over the banks
struct Fetch64 {
  const double* base;  // this thread's pair in k-tile 0
  int row, kk;         // global row of the pair's first element; k offset inside a k-tile
  int lds;             // where the pair goes in the [k][row] image
  long sRow, sK, step;
  int rows, K;
  bool kfast;
  __device__ __forceinline__ void init(const double* P, long sRow_, long sK_, int row0, int rows_, int K_, int tid) {
    sRow = sRow_; sK = sK_; rows = rows_; K = K_; kfast = sK_ == 1;
    const int e = 2 * tid;  // 64 x 8 elements, two per thread along the contiguous stride
    int r;
    if (kfast) { r = e >> 3; kk = e & 7; } else { r = e & 63; kk = e >> 6; }
    row = row0 + r;
    lds = kk * LD64 + r;
    base = P + (long)row * sRow + (long)kk * sK;
    step = (long)BK64 * sK;
  }
  __device__ __forceinline__ void load(int kt, double (&v)[2]) const {
    const double* src = base + (long)kt * step;
    const int gk = kt * BK64 + kk;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r2 = kfast ? row : row + j, k2 = kfast ? gk + j : gk;
      v[j] = (r2 < rows && k2 < K) ? src[kfast ? (long)j * sK : (long)j * sRow] : 0.0;
    }
  }
  __device__ __forceinline__ void store(double* T, const double (&v)[2]) const {
    if (kfast) { T[lds] = v[0]; T[lds + LD64] = v[1]; }
    else { T[lds] = v[0]; T[lds + 1] = v[1]; }
  }
};
__global__ void __launch_bounds__(256) k_bgemm_f64(BGT<double> g) {
  __shared__ __attribute__((aligned(16))) double As[2][BK64 * LD64];
  __shared__ __attribute__((aligned(16))) double Bs[2][BK64 * LD64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, gq = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const BlockId bid = xcd_block();
  const int m0 = bid.y * 64, n0 = bid.x * 64;
  const long b = bid.z;
  f64x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;
  Fetch64 FA, FB;
  FA.init(g.A + b * g.bA, g.sAm, g.sAk, m0, g.M, g.K, tid);
  FB.init(g.B + b * g.bB, g.sBn, g.sBk, n0, g.N, g.K, tid);
  const int ktiles = (g.K + BK64 - 1) / BK64;
  const bool do_rowsum = g.rowsum != nullptr && bid.x == 0 && tid < 64;
  double rsum = 0.0;
  double fa[2], fb[2];
  FA.load(0, fa);
  FB.load(0, fb);
  FA.store(As[0], fa);
  FB.store(Bs[0], fb);
  __syncthreads();
  for (int kt = 0; kt < ktiles; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < ktiles) { FA.load(kt + 1, fa); FB.load(kt + 1, fb); }
    const double* Ac = As[cur];
    const double* Bc = Bs[cur];
#pragma unroll
    for (int s4 = 0; s4 < BK64 / 4; ++s4) {
      double av[2], bv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        av[i] = Ac[(4 * s4 + gq) * LD64 + wm * 32 + 16 * i + c];
        bv[i] = Bc[(4 * s4 + gq) * LD64 + wn * 32 + 16 * i + c];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    if (do_rowsum) {
#pragma unroll
      for (int k = 0; k < BK64; ++k) rsum += Ac[k * LD64 + tid];
    }
    if (kt + 1 < ktiles) {
      FA.store(As[cur ^ 1], fa);
      FB.store(Bs[cur ^ 1], fb);
    }
    __syncthreads();
  }
  // ---- epilogue (the kinds of bg_epilogue, in double with the library functions)
  const double tscale = g.pr_temp ? g.pr_temp[b] : 1.0;
  if (do_rowsum && m0 + tid < g.M) {
    const int mm = m0 + tid;
    if (g.pr_theta_b) rsum = (rsum - (g.pr_theta_b[b * g.bRow + mm] - g.pr_mu_b[mm]) * g.pr_iv_b[mm]) * tscale;
    g.rowsum[b * g.bRow + mm] = rsum;
  }
  double* C = g.C + b * g.bC;
  const double* Hm = g.Hm ? g.Hm + b * g.bH : nullptr;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 32 + 16 * j + c;
    if (n >= g.N) continue;
    const double bias = g.bias ? g.bias[b * g.bBias + n] : 0.0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 32 + 16 * i + 4 * r + gq;
        if (m >= g.M) continue;
        double v = acc[i][j][r];
        if (Hm) v *= l_dact(g.act_h, Hm[m * g.sHm + n * g.sHn]);
        else v = l_act(g.act, v + bias);
        const long ci = m * g.sCm + n * g.sCn;
        if (g.pr_theta) v = (v - (g.pr_theta[b * g.bC + ci] - g.pr_mu[ci]) * g.pr_iv[ci]) * tscale;
        C[ci] = v;
      }
    }
  }
}
static int bgemm(const BGT<double>& g, int batch, hipStream_t s, int* cursor = nullptr, bool dry = false) {
  {
    bool handled;
    const int rc = bgemm_narrow(g, batch, s, cursor, dry, &handled);
    if (handled) return rc;
  }
  dim3 grid((g.N + 63) / 64, (g.M + 63) / 64, batch);
  hipLaunchKernelGGL(k_bgemm_f64, grid, dim3(256), 0, s, g);
  EY_HIP(hipGetLastError());
  return EY_OK;
}

// A thread-strided pass over n elements with the SAME trip count in every thread (see EY_LANE_PASS in ey_generic.hip): index
// i clamped to n - 1, `on` saying whether the thread's element exists.  Sums select their term's input to zero for a thread
// beyond n: no register carried across a loop whose last round runs under a partial EXEC mask.
#define EY_THREAD_PASS(n, i, on)                                                                                      \
  for (int ey_k_ = 0, ey_n_ = (n), ey_s_ = (int)blockDim.x, i = (int)threadIdx.x < ey_n_ ? (int)threadIdx.x : ey_n_ - 1, \
           on = (int)threadIdx.x < ey_n_;                                                                             \
       ey_k_ < (ey_n_ + ey_s_ - 1) / ey_s_;                                                                           \
       ++ey_k_, on = (int)threadIdx.x + ey_k_ * ey_s_ < ey_n_, i = on ? (int)threadIdx.x + ey_k_ * ey_s_ : ey_n_ - 1)

template <class T>
__device__ __forceinline__ T block_sum(T v, T* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  T t = T(0.0);
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
  return t;
}

// log-likelihood of every chain and the output delta = dL/dh_K * act'(h_K); one block per chain
template <class T>
__global__ void __launch_bounds__(256) k_loss(const T* __restrict__ out, T* __restrict__ delta,
                                              const T* __restrict__ y, const int* __restrict__ labels, int N, int dK,
                                              int lik_code, int act_last, T* __restrict__ lik_o,
                                              T* __restrict__ rows_o, const T* __restrict__ temp) {
  __shared__ T red[4];
  const long c = blockIdx.x;
  const T* o = out + c * (long)N * dK;
  T* d = delta + c * (long)N * dK;
  T lik = T(0.0);
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    T row = T(0.0);  // this row's term of the sum (ey_log_lik_rows)
    if (lik_code == EY_LIK_BCE_SUM) {
      for (int j = 0; j < dK; ++j) {
        const T p = o[n * dK + j], yy = y[n * dK + j];
        row += l_log(p) * yy + l_log(T(1.0) - p) * (T(1.0) - yy);  // naive logs (eeyore/stats/loss.py:2)
        d[n * dK + j] = (yy / p - (T(1.0) - yy) / (T(1.0) - p)) * l_dact(act_last, p);
      }
    } else {
      const int lab = labels[n];
      T mx = o[n * dK];
      for (int j = 1; j < dK; ++j) mx = l_max(mx, o[n * dK + j]);
      T ssum = T(0.0);
      for (int j = 0; j < dK; ++j) ssum += l_exp(o[n * dK + j] - mx);
      row = o[n * dK + lab] - (mx + l_log(ssum));
      const T rs = T(1.0) / ssum;
      for (int j = 0; j < dK; ++j) {
        const T v = o[n * dK + j];
        d[n * dK + j] = ((j == lab ? T(1.0) : T(0.0)) - l_exp(v - mx) * rs) * l_dact(act_last, v);
      }
    }
    lik += row;
    if (rows_o) rows_o[c * (long)N + n] = temp ? row * temp[c] : row;
  }
  lik = block_sum(lik, red);
  if (threadIdx.x == 0) lik_o[c] = lik;
}

// prior value, temperature, log-target; one block per chain
template <class T>
__global__ void __launch_bounds__(256) k_prior(const T* __restrict__ theta, const T* __restrict__ mu,
                                               const T* __restrict__ iv, T prior_const, int P,
                                               const T* __restrict__ temp, const T* __restrict__ lik,
                                               T* lik_o, T* prior_o, T* target_o) {
  __shared__ T red[4];
  const long c = blockIdx.x;
  const T t = temp ? temp[c] : T(1.0);
  T q = T(0.0);
  EY_THREAD_PASS(P, i, on) {
    const T d = on ? theta[c * P + i] - mu[i] : T(0.0);
    q += d * d * iv[i];
  }
  q = block_sum(q, red);
  if (threadIdx.x == 0) {
    const T pr = (prior_const - T(0.5) * q) * t, lk = lik[c] * t;
    if (lik_o) lik_o[c] = lk;
    if (prior_o) prior_o[c] = pr;
    if (target_o) target_o[c] = lk + pr;
  }
}

// ---- HMC elementwise pieces (one block per chain)
template <class T>
__global__ void __launch_bounds__(256) k_hmc_begin(const T* __restrict__ theta, const T* __restrict__ grad,
                                                   const T* __restrict__ p0, T* __restrict__ thp, T* __restrict__ p,
                                                   T* __restrict__ gp, int P, uint64_t seed, uint64_t iter,
                                                   uint64_t chain_offset, const T* target, T* hcur) {
  __shared__ T red[4];
  const long c = blockIdx.x;
  const EyRng rn = ey_rng_make(seed, chain_offset + (uint64_t)c, iter, EY_STREAM_NORMAL);
  T kin = T(0.0);
  // momentum (hmc.py:134): one block of four stream elements per thread and round (one Philox call each).  The arrays are
  // distinct (__restrict__) and the rounds unrolled by four, so that the loads of four rounds are in flight together: as
  // plain pointers every element's copy waited for the store before it (5.6 ms for config 5's share, 1.2 TB/s)
  // (the same number of rounds in every thread, the kinetic energy summed outside the per-element branch: DESIGN.md 4.4)
  const int nb4 = (P + 3) / 4, rounds = (nb4 + (int)blockDim.x - 1) / (int)blockDim.x;
  // (round 5: a block of four that lies inside P moves as one 16-byte piece per array -- a chain's arrays start at a multiple of
  // P elements, so the pieces are only element-aligned, as any global access on this device may be; element by element every
  // wave instruction touched a quarter of the bytes of the lines it opened: 3.1 ms for config 5's share, 2.7 TB/s)
  typedef T vec4e __attribute__((ext_vector_type(4), aligned(sizeof(T))));
#pragma unroll 4
  for (int r = 0; r < rounds; ++r) {
    const int b = (int)threadIdx.x + r * (int)blockDim.x;
    T o[4];
    if (!p0) ey_rng_normal4<T>(rn, (uint32_t)b, o);
    const bool whole = 4 * b + 3 < P;
    const long k0 = c * P + (whole ? 4 * b : 0);
    T pz[4];
    if (whole) {
      vec4e pv4;
      if (p0) pv4 = *reinterpret_cast<const vec4e*>(p0 + k0);
      else pv4 = vec4e{o[0], o[1], o[2], o[3]};
      *reinterpret_cast<vec4e*>(p + k0) = pv4;
      *reinterpret_cast<vec4e*>(thp + k0) = *reinterpret_cast<const vec4e*>(theta + k0);
      *reinterpret_cast<vec4e*>(gp + k0) = *reinterpret_cast<const vec4e*>(grad + k0);
#pragma unroll
      for (int j = 0; j < 4; ++j) pz[j] = pv4[j];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = 4 * b + j;
        const bool on = i < P;
        const long k = c * P + (on ? i : P - 1);
        const T pv = p0 ? p0[k] : o[j];
        pz[j] = on ? pv : T(0.0);
        if (on) {
          p[k] = pv;
          thp[k] = theta[k];
          gp[k] = grad[k];
        }
      }
    }
    // (summed outside the branches, element after element as before: the same bits)
#pragma unroll
    for (int j = 0; j < 4; ++j) kin += pz[j] * pz[j];
  }
  kin = block_sum(kin, red);
  if (threadIdx.x == 0) hcur[c] = -target[c] + T(0.5) * kin;  // hmc.py:91-98,137
}

// p += wp * eps * g ; then theta += wt * eps * p   (either weight may be 0)
// Also leaves, per block, the partial sum of (theta - mu)^2 / sigma^2 over the block's slice of the NEW position in
// qpart[c][blockIdx.x]: the next evaluation's log-prior then needs no pass over theta (summed in a fixed order, so the
// result is reproducible).
#define LEAP_EPT 8                 // elements per thread: eight independent load chains in flight
#define LEAP_BLOCK (256 * LEAP_EPT)
static inline int leap_blocks(int P) { return (P + LEAP_BLOCK - 1) / LEAP_BLOCK; }
template <class T>
__global__ void __launch_bounds__(256) k_leap(T* thp, T* p, const T* gp, int P, T step,
                                              const T* step_vec, T wp, T wt, const T* __restrict__ mu,
                                              const T* __restrict__ iv, T* __restrict__ qpart) {
  __shared__ T red[4];
  const long c = blockIdx.y;
  const int i0 = blockIdx.x * LEAP_BLOCK + threadIdx.x;
  const T eps = step_vec ? step_vec[c] : step;
  const T ep = wp * eps, et = wt * eps;
  T pv[LEAP_EPT], tv[LEAP_EPT], gv[LEAP_EPT], mv[LEAP_EPT], vv[LEAP_EPT];
#pragma unroll
  for (int j = 0; j < LEAP_EPT; ++j) {
    const int i = i0 + j * 256;
    const long k = c * P + i;
    const bool in = i < P;
    pv[j] = in ? p[k] : T(0.0);
    gv[j] = in && wp != T(0.0) ? gp[k] : T(0.0);
    tv[j] = in ? thp[k] : T(0.0);
    mv[j] = in ? mu[i] : T(0.0);
    vv[j] = in ? iv[i] : T(0.0);
  }
  T q = T(0.0);
#pragma unroll
  for (int j = 0; j < LEAP_EPT; ++j) {
    const int i = i0 + j * 256;
    const long k = c * P + i;
    if (wp != T(0.0)) pv[j] = pv[j] + ep * gv[j];
    if (wt != T(0.0)) tv[j] = tv[j] + et * pv[j];
    if (i < P) {
      if (wp != T(0.0)) p[k] = pv[j];
      if (wt != T(0.0)) thp[k] = tv[j];
    }
    const T d = tv[j] - mv[j];  // (an element beyond P holds zeros throughout: the sum stays outside the per-thread branch)
    q += d * d * vv[j];
  }
  q = block_sum(q, red);
  if (threadIdx.x == 0) qpart[c * gridDim.x + blockIdx.x] = q;
}

// log-target from the likelihood and the per-block partials of the prior quadratic form; one wave per chain, lanes
// stride over the partials and combine with a fixed shuffle tree (reproducible)
template <class T>
__global__ void __launch_bounds__(256) k_target(const T* __restrict__ qpart, int nblk, T prior_const,
                                                const T* __restrict__ temp, const T* __restrict__ lik, int C,
                                                T* lik_o, T* prior_o, T* target_o) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (c >= C) return;
  T q = T(0.0);
  for (int r = 0; r < (nblk + 63) / 64; ++r) {  // (the same trip count in every lane)
    const int j = lane + 64 * r;
    q += j < nblk ? qpart[(long)c * nblk + j] : T(0.0);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  if (lane != 0) return;
  const T t = temp ? temp[c] : T(1.0);
  const T pr = (prior_const - T(0.5) * q) * t, lk = lik[c] * t;
  if (lik_o) lik_o[c] = lk;
  if (prior_o) prior_o[c] = pr;
  if (target_o) target_o[c] = lk + pr;
}

template <class T>
__global__ void __launch_bounds__(256) k_hmc_end(T* __restrict__ theta, T* __restrict__ grad, T* target,
                                                 const T* __restrict__ thp, const T* __restrict__ p,
                                                 const T* __restrict__ gp, const T* tprop, const T* hcur,
                                                 const T* u_in, int P, uint64_t seed, uint64_t iter,
                                                 uint64_t chain_offset, unsigned char* accepted, T* rate_o,
                                                 T* hcur_o, T* hprop_o) {
  __shared__ T red[4];
  __shared__ int s_acc;
  const long c = blockIdx.x;
  T kin = T(0.0);
  EY_THREAD_PASS(P, i, on) {
    const T pz = on ? p[c * P + i] : T(0.0);
    kin += pz * pz;
  }
  kin = block_sum(kin, red);
  if (threadIdx.x == 0) {
    const T h_prop = -tprop[c] + T(0.5) * kin;
    T rate = l_exp(hcur[c] - h_prop);  // hmc.py:143-146
    if (rate > T(1.0)) rate = T(1.0);
    const EyRng ru = ey_rng_make(seed, chain_offset + (uint64_t)c, iter, EY_STREAM_UNIFORM);
    const T u = u_in ? u_in[c] : ey_rng_uniform<T>(ru);
    const int acc = u < rate;  // strict <, NaN => reject (hmc.py:148)
    s_acc = acc;
    accepted[c] = (unsigned char)acc;
    if (acc) target[c] = tprop[c];
    if (rate_o) rate_o[c] = rate;
    if (hcur_o) hcur_o[c] = hcur[c];
    if (hprop_o) hprop_o[c] = h_prop;
  }
  __syncthreads();
  if (s_acc) {
#pragma unroll 8
    for (int i = threadIdx.x; i < P; i += blockDim.x) {
      theta[c * P + i] = thp[c * P + i];
      grad[c * P + i] = gp[c * P + i];
    }
  }
}

// ----------------------------------------------------------------------------------------------- host
// nvec = state vectors the generic kernel of the operation in question carves from LDS (2 for a value / random-walk MH
// draw, 3 for HMC, 4 for MALA, ey_generic.hip): a model may fit for one operation and not for another
bool ey_large_needed(const ey_plan* pl, int nvec) {
  const EyModel& m = pl->m;
  const size_t Ppad = (m.P + 3) & ~3;
  const size_t esz = pl->dtype == EY_F32 ? 4 : 8;
  const size_t generic_bytes = esz * ((size_t)nvec * Ppad + (size_t)m.hrows * 65 + 2 * (size_t)m.dmax * 65);
  return generic_bytes > 160 * 1024;
}


// ---- the narrow last layer, fused (d_K <= 10 outputs, d = d_{K-1} in {16, 32, 64, 128} inputs): logits, loss, output
// delta, dW_{K-1}, db_{K-1} and delta_{K-1} = (delta_K W_{K-1}) * act'(H_{K-1}) from ONE pass over H_{K-1}.  As separate
// launches (narrow forward GEMM, k_loss, narrow dW GEMM, input-gradient product) H_{K-1} is read three times; every one
// of those is HBM-bound, so this kernel's cost is reading H_{K-1} and writing delta_{K-1} once each.
// One workgroup per chain.  Thread (rs = tid / 16, q = tid % 16) owns features F q .. F q + F - 1 of row 16 t + rs in pass
// t: a wave reads 4 consecutive rows (contiguous in memory), 3 passes ahead.  W_{K-1} (d_K x F per thread) and the
// dW accumulators stay in registers; the 16 lanes of a row combine their partial logits with DPP rotations so that
// all of them hold the row's logits and compute the (cheap) loss redundantly.  Sums over rows are combined in a fixed
// order (slots inside a wave, then waves): reproducible.
#define TAIL_DK 10
#ifndef TAIL_PF
#define TAIL_PF 3
#endif
template <class T>
struct TailArgsT {
  const T* H; T* Dout; const T* theta; T* grad; const T* mu; const T* iv;
  const T* y; const int* labels; const T* temp; T* lik_o; T* rows_o;
  long P; int woff, boff, N, d, dK, lik, act_last, act_prev, rows_temp;
  // the leapfrog update fused into the gradient write-out, as BGT's lf_* (the workgroup's slot is lf_slot0)
  T *lf_p, *lf_q;
  const T* lf_step_vec;
  T lf_step, lf_wp, lf_wt;
  int lf_slot0, lf_nslots, lf_store_g;
};
__device__ __forceinline__ float l_act_fast(int code, float g) {
  switch (code) {
    case EY_ACT_SIGMOID: return l_sigmoid_fast(g);
    case EY_ACT_TANH: return l_tanh_fast(g);
    case EY_ACT_RELU: return g > 0.0f ? g : 0.0f;
    default: return g;
  }
}
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// sum over the 16 lanes of a DPP row, result in every lane (row_ror 8, 4, 2, 1)
__device__ __forceinline__ float row16_sum(float v) {
  v = dpp_add<0x128>(v); v = dpp_add<0x124>(v); v = dpp_add<0x122>(v); v = dpp_add<0x121>(v);
  return v;
}
template <int CTRL>
__device__ __forceinline__ double dpp_add(double v) {  // the two halves of the rotated value move separately
  const long long bits = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)bits, CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(bits >> 32), CTRL, 0xf, 0xf, true);
  return v + __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double row16_sum(double v) {
  v = dpp_add<0x128>(v); v = dpp_add<0x124>(v); v = dpp_add<0x122>(v); v = dpp_add<0x121>(v);
  return v;
}
__device__ __forceinline__ double l_act_fast(int code, double g) { return l_act(code, g); }
// `room` = how many of this lane's F features exist (d - F q: the layer may be narrower than the 16 F the lanes span;
// its width is a multiple of the vector piece -- 4, 2 or 1 floats -- so a piece is whole or absent)
// No branch depends on the lane or the row: a piece that does not exist (features beyond d, rows beyond N) is loaded from
// a piece that does -- `rowp` is a row inside the batch, the absent piece reads the row's first -- and dropped where the
// values are USED (tail_mask).  Loads inside per-lane branches made the compiler wait for them where the branches merge:
// with them (and the row's label fetched in the pass that used it) every pass of k_tail waited for its own prefetches,
// 3 us per pass of 16 rows in config 5 (profiles/r04_cfg5_tail.txt).
template <int F, class T>
__device__ __forceinline__ void tail_load(const T* rowp, int q, int room, T (&h)[F]) {
  if constexpr (F >= 4) {
#pragma unroll
    for (int f = 0; f < F; f += 4) {
      const Vec4<T> v = *reinterpret_cast<const Vec4<T>*>(rowp + (f < room ? F * q + f : 0));
      h[f] = v.x; h[f + 1] = v.y; h[f + 2] = v.z; h[f + 3] = v.w;
    }
  } else if constexpr (F == 2) {
    const Vec2<T> v = *reinterpret_cast<const Vec2<T>*>(rowp + (room > 0 ? 2 * q : 0));
    h[0] = v.x; h[1] = v.y;
  } else {
    h[0] = rowp[room > 0 ? q : 0];
  }
}
template <int F, class T>
__device__ __forceinline__ void tail_mask(bool ok, int room, const T (&raw)[F], T (&h)[F]) {
#pragma unroll
  for (int f = 0; f < F; ++f) h[f] = (ok && (F >= 4 ? (f & ~3) : 0) < room) ? raw[f] : T(0.0);
}
// Unconditional: a piece that must not be written (rows beyond N, features beyond d) goes to the lane's own slot of a junk
// buffer instead.  Behind a per-lane branch the stores made the compiler merge two paths' counts of outstanding memory
// operations conservatively, and every pass of k_tail waited for most of its prefetch ring.
__device__ double g_tail_junk[256 * 8];
template <int F, class T>
__device__ __forceinline__ void tail_store(T* p, bool ok, int room, const T (&h)[F]) {
  T* junk = reinterpret_cast<T*>(g_tail_junk) + threadIdx.x * F;
  if constexpr (F >= 4) {
#pragma unroll
    for (int f = 0; f < F; f += 4) {
      T* dst = (ok && f < room) ? p + f : junk + f;
      *reinterpret_cast<Vec4<T>*>(dst) = Vec4<T>{h[f], h[f + 1], h[f + 2], h[f + 3]};
    }
  } else if constexpr (F == 2) {
    *reinterpret_cast<Vec2<T>*>((ok && room > 0) ? p : junk) = Vec2<T>{h[0], h[1]};
  } else {
    *((ok && room > 0) ? p : junk) = h[0];
  }
}
template <int F, int WS, class T>
__device__ __forceinline__ void tail_wrow(const T* wl, int j, int q, T (&w)[F]) {
  if constexpr (F >= 4) {
#pragma unroll
    for (int p = 0; p < F / 4; ++p) {
      const Vec4<T> v = *reinterpret_cast<const Vec4<T>*>(wl + (p * TAIL_DK + j) * WS + 4 * q);
      w[4 * p] = v.x; w[4 * p + 1] = v.y; w[4 * p + 2] = v.z; w[4 * p + 3] = v.w;
    }
  } else if constexpr (F == 2) {
    const Vec2<T> v = *reinterpret_cast<const Vec2<T>*>(wl + j * WS + 2 * q);
    w[0] = v.x; w[1] = v.y;
  } else {
    w[0] = wl[j * WS + q];
  }
}
// sum over the LPR lanes of a row, result in every lane of the row
template <int LPR, class T>
__device__ __forceinline__ T row_sum(T v) {
  v = row16_sum(v);
  if (LPR == 32) v += __shfl_xor(v, 16, 64);
  return v;
}
// LPR lanes per row, RPP = 256 / LPR rows per pass.  LPR = 16 is what runs: for d > 64 that means 8 features per lane, 80
// accumulators and two waves per SIMD; the alternative (32 lanes x 4 features, three waves per SIMD, W in registers)
// measured 2-6 % slower -- the loss is computed redundantly by every lane of a row and the row sums cross a DPP row.
template <class T, int F, int LPR, bool GRAD>
__global__ void __launch_bounds__(256, (F * sizeof(T) >= 32) ? 2 : 3) k_tail(TailArgsT<T> a) {
  constexpr int RPP = 256 / LPR, WS = LPR * 4;  // rows per pass; floats per row of a W plane
  __shared__ T red[4][TAIL_DK * LPR * F];
  __shared__ __attribute__((aligned(16))) T wl[(F >= 4 ? F / 4 : 1) * TAIL_DK * WS];
  __shared__ T redb[4][TAIL_DK];
  __shared__ T redl[4];
  const int tid = threadIdx.x, q = tid & (LPR - 1), rs = tid / LPR, wave = tid >> 6, lane = tid & 63;
  const long c = blockIdx.x;
  const int N = a.N, d = a.d, dK = a.dK;
  const T* Hc = a.H + c * (long)N * d;
  const int room = d - F * q;  // this lane's features that exist
  const T* th = a.theta + c * a.P;
  // W_{K-1} in LDS (rows beyond d_K zero), laid out so that the 16 lanes of a row read consecutive 16-byte pieces:
  // plane p holds features F q + 4 p .. + 3 of lane q.  (In registers it would cost 10 F of them next to the 10 F
  // accumulators; the reads are broadcasts over the wave's four rows and cost a few LDS cycles per pass.)
  {  // all of a thread's elements are fetched before any is stored: the loads overlap (this is per-chain latency)
    constexpr int NW = (TAIL_DK * LPR * F + 255) / 256;
    T wv[NW];
#pragma unroll
    for (int u = 0; u < NW; ++u) {
      const int idx = tid + 256 * u, j = idx / (LPR * F), i = idx - j * (LPR * F);
      wv[u] = (idx < TAIL_DK * LPR * F && j < dK && i < d) ? th[a.woff + j * d + i] : T(0.0);
    }
#pragma unroll
    for (int u = 0; u < NW; ++u) {
      const int idx = tid + 256 * u;
      if (idx < TAIL_DK * LPR * F) {
        const int j = idx / (LPR * F), i = idx - j * (LPR * F), qq = i / F, f = i - qq * F;
        const int dst = F >= 4 ? ((f >> 2) * TAIL_DK + j) * WS + 4 * qq + (f & 3) : j * WS + F * qq + f;
        wl[dst] = wv[u];
      }
    }
  }
  T bias[TAIL_DK];
#pragma unroll
  for (int j = 0; j < TAIL_DK; ++j) bias[j] = (j < dK && a.boff >= 0) ? th[a.boff + j] : T(0.0);
  __syncthreads();
  T acc[TAIL_DK][F], dbq = T(0.0);  // dbq: lane q of a row keeps the bias-gradient sum of output q (one register, not d_K)
#pragma unroll
  for (int j = 0; j < TAIL_DK; ++j) {
#pragma unroll
    for (int f = 0; f < F; ++f) acc[j][f] = T(0.0);
  }
  T lik = T(0.0);
  const T rowscale = a.rows_temp && a.temp ? a.temp[c] : T(1.0);
  const int passes = (N + RPP - 1) / RPP;
  T hb[TAIL_PF][F];
  int lb[TAIL_PF];  // the rows' labels travel with them
#pragma unroll
  for (int u = 0; u < TAIL_PF; ++u) {
    const int n = min(RPP * u + rs, N - 1);
    tail_load<F>(Hc + (long)n * d, q, room, hb[u]);
    lb[u] = a.labels[n];
  }
  for (int t0 = 0; t0 < passes; t0 += TAIL_PF) {
#pragma unroll
    for (int u = 0; u < TAIL_PF; ++u) {
      const int t = t0 + u;
      if (t >= passes) break;  // uniform
      const int n = RPP * t + rs;
      const bool live = n < N;
      T h[F];
      tail_mask<F>(live, room, hb[u], h);
      const int lab_pf = lb[u];
      // eight features per lane: q is made opaque per pass to keep the W reads below in the loop (hoisted they would
      // occupy 80 registers next to the 80 accumulators); with four or fewer the compiler hoists them, W lives in
      // registers and the loop has no LDS traffic
      int qv = q;
      if constexpr (F * sizeof(T) >= 32) asm volatile("" : "+v"(qv));
      {  // refill this slot with the row TAIL_PF passes ahead (the last passes fetch row N - 1 again: never used)
        const int n2 = min(n + RPP * TAIL_PF, N - 1);
        tail_load<F>(Hc + (long)n2 * d, q, room, hb[u]);
        lb[u] = a.labels[n2];
      }
      // logits: partial dot products over this lane's features, combined over the row's 16 lanes; rows of W beyond d_K
      // are zero.  The activation switches sit OUTSIDE the element loops (one uniform branch per pass, not per element).
      T z[TAIL_DK], da[TAIL_DK];
#pragma unroll
      for (int j = 0; j < TAIL_DK; ++j) {
        T wj[F], pz = T(0.0);
        tail_wrow<F, WS>(wl, j, qv, wj);
#pragma unroll
        for (int f = 0; f < F; ++f) pz += h[f] * wj[f];
        z[j] = row_sum<LPR>(pz) + bias[j];
      }
      act_vec<TAIL_DK>(a.act_last, z);
      dact_vec<TAIL_DK>(a.act_last, z, da);
      // the row's loss term and dL/dz (same arithmetic as k_loss), in every lane of the row
      T row = T(0.0), dl[TAIL_DK];
      if (a.lik == EY_LIK_BCE_SUM) {
#pragma unroll
        for (int j = 0; j < TAIL_DK; ++j) {
          dl[j] = T(0.0);
          if (j < dK && live) {
            const T p = z[j], yy = a.y[(long)n * dK + j];
            row += l_log(p) * yy + l_log(T(1.0) - p) * (T(1.0) - yy);  // naive logs (eeyore/stats/loss.py:2)
            dl[j] = (yy / p - (T(1.0) - yy) / (T(1.0) - p)) * da[j];
          }
        }
      } else {
        const int lab = live ? lab_pf : 0;
        T mx = z[0];
#pragma unroll
        for (int j = 1; j < TAIL_DK; ++j) mx = j < dK ? l_max(mx, z[j]) : mx;
        T e[TAIL_DK], ssum = T(0.0), zlab = T(0.0);
#pragma unroll
        for (int j = 0; j < TAIL_DK; ++j) {
          e[j] = j < dK ? l_exp(z[j] - mx) : T(0.0);
          ssum += e[j];
          zlab = j == lab ? z[j] : zlab;
        }
        row = zlab - (mx + l_log(ssum));
        const T rsum = live ? T(1.0) / ssum : T(0.0);
#pragma unroll
        for (int j = 0; j < TAIL_DK; ++j) dl[j] = ((j == lab && live ? T(1.0) : T(0.0)) - e[j] * rsum) * da[j];
      }
      lik += live ? row : T(0.0);
      if (a.rows_o) {  // (uniform: ey_log_lik_rows only)
        if (live && q == 0) a.rows_o[c * (long)N + n] = row * rowscale;
      }
      if (GRAD) {
        T dh[F];
#pragma unroll
        for (int f = 0; f < F; ++f) dh[f] = T(0.0);
#pragma unroll
        for (int j = 0; j < TAIL_DK; ++j) {
          T wj[F];
          tail_wrow<F, WS>(wl, j, qv, wj);
          dbq += j == q ? dl[j] : T(0.0);
#pragma unroll
          for (int f = 0; f < F; ++f) {
            acc[j][f] += dl[j] * h[f];
            dh[f] += dl[j] * wj[f];
          }
        }
        if (a.Dout) {  // (uniform)
          T dp[F];
          dact_vec<F>(a.act_prev, h, dp);
#pragma unroll
          for (int f = 0; f < F; ++f) dh[f] *= dp[f];
          tail_store<F>(a.Dout + c * (long)N * d + (long)n * d + F * q, live, room, dh);
        }
      }
    }
  }
  // ---- reductions over rows: slots of a wave (lane bits 4, 5), then the four waves, in that order
  T lv = q == 0 ? lik : T(0.0);
  if (LPR == 16) lv += __shfl_xor(lv, 16, 64);
  lv += __shfl_xor(lv, 32, 64);
  if (lane == 0) redl[wave] = lv;
  if (GRAD) {
    {
      T b = dbq;  // lanes q, q + 16, q + 32, q + 48: the wave's four rows
      if (LPR == 16) b += __shfl_xor(b, 16, 64);
      b += __shfl_xor(b, 32, 64);
      if (lane < TAIL_DK) redb[wave][lane] = b;
    }
#pragma unroll
    for (int j = 0; j < TAIL_DK; ++j) {
#pragma unroll
      for (int f = 0; f < F; ++f) {
        T v = acc[j][f];
        if (LPR == 16) v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (lane < LPR) red[wave][j * (LPR * F) + F * q + f] = v;
      }
    }
  }
  __syncthreads();
  if (tid == 0) a.lik_o[c] = ((redl[0] + redl[1]) + redl[2]) + redl[3];
  if (GRAD) {
    const T tscale = a.temp ? a.temp[c] : T(1.0);
    T* gc = a.grad + c * a.P;
    const bool fuse = a.lf_p != nullptr;
    T* thw = const_cast<T*>(th);
    T* pc = fuse ? a.lf_p + c * a.P : nullptr;
    const T eps = fuse ? (a.lf_step_vec ? a.lf_step_vec[c] : a.lf_step) : T(0.0);
    const T ep = a.lf_wp * eps, et = a.lf_wt * eps;
    T q = T(0.0);
    // one gradient element: the prior term and the temperature; with the leapfrog update fused in (see BGT) the
    // momentum and the position of the element move here and q collects the new position's prior quadratic form
    auto emit = [&](int k, T v, T m_, T i_, T tv, T pin) {
      const T gv = (v - (tv - m_) * i_) * tscale;
      if (!fuse) { gc[k] = gv; return; }
      if (a.lf_store_g) gc[k] = gv;
      const T pv = pin + ep * gv;
      pc[k] = pv;
      if (a.lf_wt != T(0.0)) { tv = tv + et * pv; thw[k] = tv; }
      const T dd = tv - m_;
      q += dd * dd * i_;
    };
    // the operands of all of a thread's elements are fetched before the first is written (per-chain latency again)
    constexpr int NE = (TAIL_DK * LPR * F + 255) / 256;
    T em[NE], ei[NE], et_[NE], epn[NE];
#pragma unroll
    for (int u = 0; u < NE; ++u) {
      const int e = tid + 256 * u, k = a.woff + e;
      const bool in = e < dK * d;
      em[u] = in ? a.mu[k] : T(0.0); ei[u] = in ? a.iv[k] : T(0.0); et_[u] = in ? th[k] : T(0.0);
      epn[u] = in && fuse ? pc[k] : T(0.0);
    }
#pragma unroll
    for (int u = 0; u < NE; ++u) {
      const int e = tid + 256 * u;
      if (e < dK * d) {
        const int j = e / d, i = e - j * d;
        emit(a.woff + e, ((red[0][j * (LPR * F) + i] + red[1][j * (LPR * F) + i]) + red[2][j * (LPR * F) + i]) +
                             red[3][j * (LPR * F) + i], em[u], ei[u], et_[u], epn[u]);
      }
    }
    if (a.boff >= 0 && tid < dK) {
      const int k = a.boff + tid;
      emit(k, ((redb[0][tid] + redb[1][tid]) + redb[2][tid]) + redb[3][tid], a.mu[k], a.iv[k], th[k],
           fuse ? pc[k] : T(0.0));
    }
    if (fuse) {  // uniform over the workgroup
      __shared__ T redq[4];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
      if (lane == 0) redq[wave] = q;
      __syncthreads();
      if (tid == 0) a.lf_q[c * a.lf_nslots + a.lf_slot0] = ((redq[0] + redq[1]) + redq[2]) + redq[3];
    }
  }
}
// d_{K-1} <= 128 and a multiple of the lanes' vector piece: the 16 lanes of a row span 16 F >= d features
// (f64 beyond d = 64 takes 32 lanes x 4 features per row: eight doubles per lane would be 160 accumulator registers)
static int tail_f(int d) { return d <= 16 ? 1 : (d <= 32 ? 2 : 4); }  // vector piece: features that must come whole
static bool tail_ok(const EyModel& m) {
  const int K = m.nl, d = m.dims[K - 1];
  if (K < 2 || m.dims[K] > TAIL_DK || d < 1 || d > 128) return false;
  return d % tail_f(d) == 0;
}
template <class T, bool GRAD>
static void tail_launch(const TailArgsT<T>& a, int C, hipStream_t s) {
  if (a.d <= 16) hipLaunchKernelGGL((k_tail<T, 1, 16, GRAD>), dim3(C), dim3(256), 0, s, a);
  else if (a.d <= 32) hipLaunchKernelGGL((k_tail<T, 2, 16, GRAD>), dim3(C), dim3(256), 0, s, a);
  else if (a.d <= 64) hipLaunchKernelGGL((k_tail<T, 4, 16, GRAD>), dim3(C), dim3(256), 0, s, a);
  else if constexpr (sizeof(T) == 4) hipLaunchKernelGGL((k_tail<T, 8, 16, GRAD>), dim3(C), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((k_tail<T, 4, 32, GRAD>), dim3(C), dim3(256), 0, s, a);
}
// variant bit 6: the last layer as separate launches (A/B, tests)

void ey_large_free(ey_plan* pl) {
  (void)hipFree(pl->d_work);
  pl->d_work = nullptr;
  pl->work_bytes = 0;
  (void)hipFree(pl->d_xpre);
  pl->d_xpre = nullptr;
  pl->xpre_bytes = 0;
  if (pl->xpre_event) (void)hipEventDestroy(pl->xpre_event);
  pl->xpre_event = nullptr;
}

// The data matrix x [N, d0] as the first layer's forward product takes it in the bf16x3 form (its A operand: rows n, k =
// input), split once per batch (ey_plan_set_data counts the batches) on the caller's stream: 4.8 MB for config 5's
// 1024 x 784; every workgroup of every chain then copies its pieces instead of splitting the same numbers again.
// Round 5: beside it the same matrix as the first layer's weight gradient takes it (its B operand: rows = inputs, k = n) when
// that product is a 128-wide one (more than 32 inputs): *out_b, else null.
static bool xpre_wants_b(const EyModel& m) { return m.dims[0] > 32; }
static int ensure_xpre(ey_plan* pl, hipStream_t s, const void** out, const void** out_b) {
  const EyModel& m = pl->m;
  const int N = m.N, d0 = m.dims[0];
  const int ktiles = (d0 + BK3 - 1) / BK3, ktiles_b = (N + BK3 - 1) / BK3;
  const size_t bytes_a = (size_t)16 * ktiles * 3 * N * 2;
  const size_t bytes = bytes_a + (xpre_wants_b(m) ? (size_t)16 * ktiles_b * 3 * d0 * 2 : 0);
  if (pl->xpre_bytes < bytes) {
    EY_HIP(hipDeviceSynchronize());
    (void)hipFree(pl->d_xpre);
    pl->d_xpre = nullptr;
    pl->xpre_bytes = 0;
    EY_HIP(hipMalloc(&pl->d_xpre, bytes));
    pl->xpre_bytes = bytes;
    pl->xpre_version = ~0ull;
  }
  if (pl->xpre_version != pl->data_version) {
    const long tasks = (long)ktiles * N * 2;
    hipLaunchKernelGGL(k_bf3_presplit, dim3((unsigned)((tasks + 255) / 256)), dim3(256), 0, s, (const float*)m.x, (long)d0, 1L, N,
                       d0, (u32x4_t*)pl->d_xpre);
    if (xpre_wants_b(m)) {
      const long tasks_b = (long)ktiles_b * d0 * 2;
      hipLaunchKernelGGL(k_bf3_presplit, dim3((unsigned)((tasks_b + 255) / 256)), dim3(256), 0, s, (const float*)m.x, 1L, (long)d0,
                         d0, N, (u32x4_t*)((char*)pl->d_xpre + bytes_a));
    }
    EY_HIP(hipGetLastError());
    if (!pl->xpre_event) EY_HIP(hipEventCreateWithFlags(&pl->xpre_event, hipEventDisableTiming));
    EY_HIP(hipEventRecord(pl->xpre_event, s));
    pl->xpre_stream = s;
    pl->xpre_version = pl->data_version;
  } else if (pl->xpre_event && s != pl->xpre_stream) {
    // the image was split on another stream (the first evaluation of this batch): order this stream behind it
    EY_HIP(hipStreamWaitEvent(s, pl->xpre_event, 0));
  }
  *out = pl->d_xpre;
  *out_b = xpre_wants_b(m) ? (const void*)((const char*)pl->d_xpre + bytes_a) : nullptr;
  return EY_OK;
}

static int ensure_work(ey_plan* pl, size_t bytes) {
  if (pl->work_bytes >= bytes) return EY_OK;
  EY_HIP(hipDeviceSynchronize());
  (void)hipFree(pl->d_work);
  pl->d_work = nullptr;
  pl->work_bytes = 0;
  EY_HIP(hipMalloc(&pl->d_work, bytes));
  pl->work_bytes = bytes;
  return EY_OK;
}

static size_t act_floats_per_chain(const EyModel& m) {
  size_t f = 0;
  for (int l = 1; l <= m.nl; ++l) f += (size_t)m.N * m.dims[l];
  return f;
}

// The leapfrog update of an HMC trajectory applied by the kernels that produce the gradient (f32): momentum p, step,
// weights of the momentum and position updates (hmc.py:105-119), where the partial sums of the new position's prior
// quadratic form go (q_out[c][nslots]) and whether the gradient itself is still stored (the last step: hmc.py:150).
template <class T>
struct LeapFuse {
  T* p; const T* step_vec; T step, wp, wt; T* q_out; int nslots, store_g;
};
// slots the fused update of one evaluation writes per chain (the same dispatch logic as the launches, dry)
static int leap_fuse_slots(const EyModel& m, bool tail) {
  int cursor = tail ? 1 : 0;
  for (int l = (tail ? m.nl - 2 : m.nl - 1); l >= 0; --l) {
    BG g = {};
    // the fields the dispatcher looks at, as eval_chunk sets them for a weight-gradient product
    g.M = m.dims[l + 1]; g.N = m.dims[l]; g.K = m.N; g.sCm = m.dims[l]; g.sCn = 1;
    g.sAm = 1; g.sAk = m.dims[l + 1]; g.sBk = m.dims[l]; g.sBn = 1;
    g.pr_theta = (const float*)m.mu; g.pr_theta_b = m.boff[l] >= 0 ? (const float*)m.mu : nullptr;
    g.rowsum = m.boff[l] >= 0 ? (float*)m.mu : nullptr;  // never dereferenced in a dry run
    bgemm(g, 1, nullptr, &cursor, true);
  }
  return cursor;
}
// variant bit 7: HMC with the separate leapfrog kernel (A/B, tests)

// value (+ gradient when grad != null) for chains [0, C) of theta, using `ws` (2 * C * act_floats floats) as scratch;
// qpart / nblk: partial sums of the prior quadratic form of theta left by the previous leapfrog update (else k_prior)
template <class T>
static int eval_chunk(ey_plan* pl, const T* theta, const T* temp, int C, T* lik_o, T* prior_o,
                      T* target_o, T* grad, T* ws, T* lik_tmp, hipStream_t s,
                      const T* qpart = nullptr, T* rows_o = nullptr, int nblk = 0, const LeapFuse<T>* lf = nullptr) {
  const EyModel& m = pl->m;
  const int K = m.nl, N = m.N, P = m.P;
  const size_t af = act_floats_per_chain(m);
  T* Hbase = ws;
  T* Dbase = ws + (size_t)C * af;
  std::vector<T*> H(K + 1), D(K + 1);
  {
    size_t off = 0;
    for (int l = 1; l <= K; ++l) {
      H[l] = Hbase + off * C;
      D[l] = Dbase + off * C;
      off += (size_t)N * m.dims[l];
    }
  }
  int rc, cursor = 0;
  // bf16x3 form, first layer wide enough for the 128-wide product: x comes pre-split (variant bit 11 switches it off)
  const void* xpre = nullptr;
  const void* xpreT = nullptr;  // ... and as the B operand of the first layer's weight gradient
  if constexpr (sizeof(T) == 4) {
    if (t_ey_products == EY_PRODUCTS_BF16X3 && N > 32 && m.dims[1] > 32 && m.dims[0] >= 16 && !EY_VBIT(11))
      if ((rc = ensure_xpre(pl, s, &xpre, &xpreT))) return rc;
  }
  // f32 only: in f64 the fused kernel measured SLOWER than the separate launches (every lane of a row repeats the
  // row's loss with the library exp / log, and eight-byte accumulators spill) -- 1.0 ms against 0.6 ms on MLP(10-100-10)
  const bool tail = sizeof(T) == 4 && tail_ok(m) && !EY_VBIT(6);
  // mid-size models (ey_mid.hip, variant bit 13): value and gradient in ONE launch, a workgroup per chain with the weights
  // resident in LDS.  Opt-in: measured level with the launches below on three-layer models (1.00 - 1.09 x) and behind
  // them on two-layer ones (0.67 x), profiles/r05_mid_ab.txt
  bool mid = false;
  if constexpr (sizeof(T) == 4) {
    if (grad && !rows_o && !lf && EY_VBIT(13) && ey_mid_supports(pl)) {
      if ((rc = ey_mid_eval(pl, (const float*)theta, (const float*)temp, C, (float*)lik_tmp, (float*)grad, s))) return rc;
      mid = true;
    } else if (grad && !rows_o && !lf && !EY_VBIT(14) && ey_mid32_supports(pl)) {
      // narrow deeper models (every hidden width <= 32): one wave per row tile, no barrier inside a chain's rounds
      if ((rc = ey_mid32_eval(pl, (const float*)theta, (const float*)temp, C, (float*)lik_tmp, (float*)grad, (void*)ws, s))) return rc;
      mid = true;
    }
  }
  if (!mid) {
  for (int l = 0; l < (tail ? K - 1 : K); ++l) {
    BGT<T> g = {};
    g.A = l == 0 ? (const T*)m.x : H[l];
    g.B = theta + m.woff[l];
    g.C = H[l + 1];
    g.M = N; g.N = m.dims[l + 1]; g.K = m.dims[l];
    g.sAm = m.dims[l]; g.sAk = 1; g.bA = l == 0 ? 0 : (long)N * m.dims[l];
    g.sBk = 1; g.sBn = m.dims[l]; g.bB = P;
    g.sCm = m.dims[l + 1]; g.sCn = 1; g.bC = (long)N * m.dims[l + 1];
    g.bias = m.boff[l] >= 0 ? theta + m.boff[l] : nullptr; g.bBias = P;
    g.act = m.act[l];
    if constexpr (sizeof(T) == 4) {
      if (l == 0 && xpre) { g.pre = xpre; g.pre_rows = N; }
    }
    if ((rc = bgemm(g, C, s))) return rc;
  }
  if constexpr (sizeof(T) == 4) {  // (the kernel is written for both types; only float is instantiated, see above)
    if (tail) {
      TailArgsT<T> t = {};
      t.H = H[K - 1]; t.Dout = grad ? D[K - 1] : nullptr; t.theta = theta; t.grad = grad;
      t.mu = (const T*)m.mu; t.iv = (const T*)m.inv_var; t.y = (const T*)m.y; t.labels = m.labels;
      t.temp = temp; t.lik_o = lik_tmp; t.rows_o = rows_o; t.P = P; t.woff = m.woff[K - 1]; t.boff = m.boff[K - 1];
      t.N = N; t.d = m.dims[K - 1]; t.dK = m.dims[K]; t.lik = m.lik; t.act_last = m.act[K - 1];
      t.act_prev = m.act[K - 2]; t.rows_temp = rows_o != nullptr;
      if (lf) {
        t.lf_p = lf->p; t.lf_q = lf->q_out; t.lf_step_vec = lf->step_vec; t.lf_step = lf->step; t.lf_wp = lf->wp;
        t.lf_wt = lf->wt; t.lf_slot0 = cursor++; t.lf_nslots = lf->nslots; t.lf_store_g = lf->store_g;
      }
      if (grad) tail_launch<T, true>(t, C, s);
      else tail_launch<T, false>(t, C, s);
    }
  }
  if (!tail)
    hipLaunchKernelGGL((k_loss<T>), dim3(C), dim3(256), 0, s, (const T*)H[K], D[K], (const T*)m.y, m.labels, N,
                       m.dims[K], m.lik, m.act[K - 1], lik_tmp, rows_o, rows_o ? temp : nullptr);
  const int ltop = tail ? K - 2 : K - 1;
  if (grad) {
    for (int l = ltop; l >= 0; --l) {
      if (l > 0) {  // first, while W_l is still the evaluated position (the fused update below moves it)
        BGT<T> d = {};  // delta_l = (delta_{l+1} W_l) * act'(H_l)
        d.A = D[l + 1]; d.sAm = m.dims[l + 1]; d.sAk = 1; d.bA = (long)N * m.dims[l + 1];
        d.B = theta + m.woff[l]; d.sBk = m.dims[l]; d.sBn = 1; d.bB = P;
        d.C = D[l]; d.sCm = m.dims[l]; d.sCn = 1; d.bC = (long)N * m.dims[l];
        d.M = N; d.N = m.dims[l]; d.K = m.dims[l + 1];
        d.Hm = H[l]; d.sHm = m.dims[l]; d.sHn = 1; d.bH = (long)N * m.dims[l]; d.act_h = m.act[l - 1];
        d.epi_nobatch = EY_VBIT(12);
        if ((rc = bgemm(d, C, s))) return rc;
      }
      BGT<T> g = {};  // dW_l = delta_{l+1}^T H_l, db_l = row sums of delta_{l+1}^T
      g.A = D[l + 1]; g.sAm = 1; g.sAk = m.dims[l + 1]; g.bA = (long)N * m.dims[l + 1];
      g.B = l == 0 ? (const T*)m.x : H[l]; g.sBk = m.dims[l]; g.sBn = 1; g.bB = l == 0 ? 0 : (long)N * m.dims[l];
      g.C = grad + m.woff[l]; g.sCm = m.dims[l]; g.sCn = 1; g.bC = P;
      g.M = m.dims[l + 1]; g.N = m.dims[l]; g.K = N; g.act = EY_ACT_NONE;
      if constexpr (sizeof(T) == 4) {
        if (l == 0 && xpreT) { g.preB = xpreT; g.preB_rows = m.dims[0]; }
      }
      if (m.boff[l] >= 0) { g.rowsum = grad + m.boff[l]; g.bRow = P; }
      g.pr_theta = theta + m.woff[l]; g.pr_mu = (const T*)m.mu + m.woff[l]; g.pr_iv = (const T*)m.inv_var + m.woff[l];
      if (m.boff[l] >= 0) {
        g.pr_theta_b = theta + m.boff[l]; g.pr_mu_b = (const T*)m.mu + m.boff[l];
        g.pr_iv_b = (const T*)m.inv_var + m.boff[l];
      }
      g.pr_temp = temp;
      if (pl->prior_uniform) { g.pr_uniform = 1; g.pr_mu0 = (T)pl->prior_mu0; g.pr_iv0 = (T)pl->prior_iv0; }
      g.epi_nobatch = EY_VBIT(12);
      if (lf) {
        g.lf_p = lf->p + m.woff[l]; g.lf_p_b = m.boff[l] >= 0 ? lf->p + m.boff[l] : nullptr; g.lf_q = lf->q_out;
        g.lf_step_vec = lf->step_vec; g.lf_step = lf->step; g.lf_wp = lf->wp; g.lf_wt = lf->wt;
        g.lf_nslots = lf->nslots; g.lf_store_g = lf->store_g;
      }
      if ((rc = bgemm(g, C, s, &cursor))) return rc;
    }
  }
  }  // !mid
  if (qpart)
    hipLaunchKernelGGL((k_target<T>), dim3((C + 3) / 4), dim3(256), 0, s, qpart, nblk ? nblk : leap_blocks(P), (T)m.prior_const,
                       temp, (const T*)lik_tmp, C, lik_o, prior_o, target_o);
  else
    hipLaunchKernelGGL((k_prior<T>), dim3(C), dim3(256), 0, s, theta, (const T*)m.mu, (const T*)m.inv_var,
                       (T)m.prior_const, P, temp, (const T*)lik_tmp, lik_o, prior_o, target_o);
  EY_HIP(hipGetLastError());
  return EY_OK;
}

static int chunk_size(const ey_plan* pl, int64_t C) {
  const size_t per_chain = 2 * act_floats_per_chain(pl->m) * (pl->dtype == EY_F32 ? 4 : 8);
  size_t cap = (size_t)16 << 30;  // activation scratch budget (of 288 GB): config 5's 4096 chains x 1024 rows are one chunk (4.6 GB)
  int64_t cc = (int64_t)(cap / (per_chain ? per_chain : 1));
  if (cc < 1) cc = 1;
  if (cc > 32768) cc = 32768;  // gridDim.z
  return (int)(cc < C ? cc : C);
}

template <class T>
static int large_log_target(ey_plan* pl, const void* theta, const void* temp, int64_t C, void* lik, void* prior,
                        void* target, void* grad, hipStream_t s) {
  const EyModel& m = pl->m;
  const int cc = chunk_size(pl, C);
  const size_t ws_floats = 2 * (size_t)cc * act_floats_per_chain(m) + (size_t)cc;
  int rc = ensure_work(pl, ws_floats * sizeof(T));
  if (rc) return rc;
  T* ws = (T*)pl->d_work;
  T* lik_tmp = ws + 2 * (size_t)cc * act_floats_per_chain(m);
  for (int64_t c0 = 0; c0 < C; c0 += cc) {
    const int n = (int)((C - c0) < cc ? (C - c0) : cc);
    rc = eval_chunk<T>(pl, (const T*)theta + c0 * m.P, temp ? (const T*)temp + c0 : nullptr, n,
                    lik ? (T*)lik + c0 : nullptr, prior ? (T*)prior + c0 : nullptr,
                    target ? (T*)target + c0 : nullptr, grad ? (T*)grad + c0 * m.P : nullptr, ws, lik_tmp, s);
    if (rc) return rc;
  }
  return EY_OK;
}
int ey_large_log_target(ey_plan* pl, const void* theta, const void* temp, int64_t C, void* lik, void* prior,
                        void* target, void* grad, hipStream_t s) {
  return pl->dtype == EY_F32 ? large_log_target<float>(pl, theta, temp, C, lik, prior, target, grad, s) : large_log_target<double>(pl, theta, temp, C, lik, prior, target, grad, s);
}

template <class T>
static int large_hmc(ey_plan* pl, void* theta, void* target, void* grad, const void* p0, const void* u, double step,
                 const void* step_vec, int L, const void* temp, int64_t C, uint64_t seed, uint64_t iter,
                 uint64_t chain_offset, uint32_t flags, void* accepted, void* rate, void* hcur, void* hprop,
                 hipStream_t s) {
  const EyModel& m = pl->m;
  const int P = m.P;
  const int cc = chunk_size(pl, C);
  const size_t af = act_floats_per_chain(m);
  // workspace: activations for a chunk + lik + [thp, p, gp] for the chunk + tprop, hcur + two buffers of partial sums
  // of the prior quadratic form (one being read by the evaluation at theta_k while its epilogues fill the other for
  // theta_{k+1})
  const int nblk = leap_blocks(P);
  // f32 only: in f64 the fused kernel measured SLOWER than the separate launches (every lane of a row repeats the
  // row's loss with the library exp / log, and eight-byte accumulators spill) -- 1.0 ms against 0.6 ms on MLP(10-100-10)
  const bool tail = sizeof(T) == 4 && tail_ok(m) && !EY_VBIT(6);
  // (the fused mid-size kernel produces the whole gradient in one launch: the leapfrog update then stays k_leap's)
  const bool fuse = sizeof(T) == 4 && !EY_VBIT(7) && !(EY_VBIT(13) && ey_mid_supports(pl)) && !(!EY_VBIT(14) && ey_mid32_supports(pl));
  const int nslots = fuse ? leap_fuse_slots(m, tail) : 0;
  const int nq = nblk > nslots ? nblk : nslots;
  const size_t ws_floats = 2 * (size_t)cc * af + (size_t)cc + 3 * (size_t)cc * P + 2 * (size_t)cc + 2 * (size_t)cc * nq;
  int rc = ensure_work(pl, ws_floats * sizeof(T));
  if (rc) return rc;
  T* ws = (T*)pl->d_work;
  T* lik_tmp = ws + 2 * (size_t)cc * af;
  T* thp = lik_tmp + cc;
  T* p = thp + (size_t)cc * P;
  T* gp = p + (size_t)cc * P;
  T* tprop = gp + (size_t)cc * P;
  T* hc = tprop + cc;
  T* qbuf[2] = {hc + cc, hc + cc + (size_t)cc * nq};
  const T* mu = (const T*)m.mu;
  const T* iv = (const T*)m.inv_var;
  for (int64_t c0 = 0; c0 < C; c0 += cc) {
    const int n = (int)((C - c0) < cc ? (C - c0) : cc);
    T* th_c = (T*)theta + c0 * P;
    T* g_c = (T*)grad + c0 * P;
    T* t_c = (T*)target + c0;
    const T* temp_c = temp ? (const T*)temp + c0 : nullptr;
    const T* sv_c = step_vec ? (const T*)step_vec + c0 : nullptr;
    hipLaunchKernelGGL((k_hmc_begin<T>), dim3(n), dim3(256), 0, s, (const T*)th_c, (const T*)g_c,
                       p0 ? (const T*)p0 + c0 * P : nullptr, thp, p, gp, P, seed, iter, chain_offset + (uint64_t)c0,
                       (const T*)t_c, hc);
    if (flags & EY_RECOMPUTE_INITIAL_GRAD) {  // hmc.py:104
      if ((rc = eval_chunk<T>(pl, thp, temp_c, n, nullptr, nullptr, tprop, gp, ws, lik_tmp, s))) return rc;
    }
    const dim3 grid(nblk, n);
    // p += eps/2 g ; theta += eps p      (hmc.py:105,110)
    hipLaunchKernelGGL((k_leap<T>), grid, dim3(256), 0, s, thp, p, (const T*)gp, P, (T)step, sv_c, T(0.5), T(1.0), mu, iv,
                       qbuf[0]);
    for (int k = 1; k <= L; ++k) {
      // full momentum step + position step, or the closing half momentum step (hmc.py:113-119): applied by the
      // kernels that produce the gradient (fuse), else by k_leap after the evaluation
      const T wp = k < L ? T(1.0) : T(0.5), wt = k < L ? T(1.0) : T(0.0);
      if (fuse) {
        const LeapFuse<T> lf = {p, sv_c, (T)step, wp, wt, qbuf[k & 1], nslots, k == L};
        if ((rc = eval_chunk<T>(pl, thp, temp_c, n, nullptr, nullptr, tprop, gp, ws, lik_tmp, s, qbuf[(k - 1) & 1], nullptr,
                                k == 1 ? nblk : nslots, &lf)))
          return rc;
      } else {
        if ((rc = eval_chunk<T>(pl, thp, temp_c, n, nullptr, nullptr, tprop, gp, ws, lik_tmp, s, qbuf[0]))) return rc;
        hipLaunchKernelGGL((k_leap<T>), grid, dim3(256), 0, s, thp, p, (const T*)gp, P, (T)step, sv_c, wp, wt, mu, iv,
                           qbuf[0]);
      }
    }
    hipLaunchKernelGGL((k_hmc_end<T>), dim3(n), dim3(256), 0, s, th_c, g_c, t_c, (const T*)thp, (const T*)p,
                       (const T*)gp, (const T*)tprop, (const T*)hc, u ? (const T*)u + c0 : nullptr, P,
                       seed, iter, chain_offset + (uint64_t)c0, (unsigned char*)accepted + c0,
                       rate ? (T*)rate + c0 : nullptr, hcur ? (T*)hcur + c0 : nullptr,
                       hprop ? (T*)hprop + c0 : nullptr);
  }
  EY_HIP(hipGetLastError());
  return EY_OK;
}
int ey_large_hmc(ey_plan* pl, void* theta, void* target, void* grad, const void* p0, const void* u, double step,
                 const void* step_vec, int L, const void* temp, int64_t C, uint64_t seed, uint64_t iter,
                 uint64_t chain_offset, uint32_t flags, void* accepted, void* rate, void* hcur, void* hprop,
                 hipStream_t s) {
  return pl->dtype == EY_F32 ? large_hmc<float>(pl, theta, target, grad, p0, u, step, step_vec, L, temp, C, seed, iter, chain_offset, flags, accepted, rate, hcur, hprop, s) : large_hmc<double>(pl, theta, target, grad, p0, u, step, step_vec, L, temp, C, seed, iter, chain_offset, flags, accepted, rate, hcur, hprop, s);
}

// ----------------------------------------------------------------------------------------------- MALA / MH / leapfrog
// proposal of every chain: MALA  prop = theta + eps/2 grad + sqrt(eps) z   (mala.py:35-41,53)
//                          MH    prop = theta + scale z                    (metropolis_hastings.py:45, normal_kernel.py)
template <class T>
__global__ void __launch_bounds__(256) k_propose(const T* __restrict__ theta, const T* __restrict__ grad,
                                                 const T* __restrict__ z_in, const T* __restrict__ scale,
                                                 T* __restrict__ prop, int P, T step,
                                                 const T* __restrict__ step_vec, T sqrt_step, uint64_t seed,
                                                 uint64_t iter, uint64_t chain_offset) {
  const long c = blockIdx.x;
  const EyRng rn = ey_rng_make(seed, chain_offset + (uint64_t)c, iter, EY_STREAM_NORMAL);
  const T eps = step_vec ? step_vec[c] : step;
  const T sc = step_vec ? l_sqrt(eps) : sqrt_step;
  for (int b = threadIdx.x; 4 * b < P; b += blockDim.x) {
    T o[4];
    if (!z_in) ey_rng_normal4<T>(rn, (uint32_t)b, o);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = 4 * b + j;
      if (i >= P) break;
      const long k = c * P + i;
      const T zi = z_in ? z_in[k] : o[j];
      prop[k] = grad ? (theta[k] + T(0.5) * eps * grad[k]) + sc * zi : theta[k] + scale[i] * zi;
    }
  }
}

// log-rate, accept and state update of every chain (mala.py:55-82; metropolis_hastings.py:47-73)
template <class T>
__global__ void __launch_bounds__(256) k_mh_finish(T* theta, T* grad, T* target, const T* prop,
                                                   const T* gprop, const T* tprop, const T* u_in, int P,
                                                   T step, const T* step_vec, T sqrt_step, uint64_t seed,
                                                   uint64_t iter, uint64_t chain_offset, unsigned char* accepted,
                                                   T* log_rate_o) {
  __shared__ T red[4];
  __shared__ int s_acc;
  const long c = blockIdx.x;
  T qf = T(0.0), qb = T(0.0);
  const T eps = step_vec ? step_vec[c] : step;
  if (gprop) {  // MALA: forward and backward proposal densities (their normalising terms cancel, mala.py:58-64)
    for (int i = threadIdx.x; i < P; i += blockDim.x) {
      const long k = c * P + i;
      const T df = prop[k] - (theta[k] + T(0.5) * eps * grad[k]);
      const T db = theta[k] - (prop[k] + T(0.5) * eps * gprop[k]);
      qf += df * df;
      qb += db * db;
    }
    qf = block_sum(qf, red);
    qb = block_sum(qb, red);
  }
  if (threadIdx.x == 0) {
    T log_rate = tprop[c] - target[c];
    if (gprop) {
      const T sc = step_vec ? l_sqrt(eps) : sqrt_step;
      log_rate += (qf - qb) * (T(1.0) / (T(2.0) * sc * sc));
    }
    const EyRng ru = ey_rng_make(seed, chain_offset + (uint64_t)c, iter, EY_STREAM_UNIFORM);
    const T u = u_in ? u_in[c] : ey_rng_uniform<T>(ru);
    const int acc = l_log(u) < log_rate;  // mala.py:66, metropolis_hastings.py:56
    s_acc = acc;
    accepted[c] = (unsigned char)acc;
    if (acc) target[c] = tprop[c];
    if (log_rate_o) log_rate_o[c] = log_rate;
  }
  __syncthreads();
  if (s_acc) {
    for (int i = threadIdx.x; i < P; i += blockDim.x) {
      theta[c * P + i] = prop[c * P + i];
      if (gprop) grad[c * P + i] = gprop[c * P + i];
    }
  }
}

template <class T>
__global__ void __launch_bounds__(256) k_negate(T* p, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = -p[i];
}

// One MALA.draw (grad != null) or MetropolisHastings.draw (scale != null) for C chains
template <class T>
static int large_mala_mh(ey_plan* pl, void* theta, void* target, void* grad, const void* z, const void* u, double step,
                     const void* step_vec, const void* scale, const void* temp, int64_t C, uint64_t seed, uint64_t iter,
                     uint64_t chain_offset, void* accepted, void* log_rate, hipStream_t s) {
  const EyModel& m = pl->m;
  const int P = m.P;
  const bool mala = grad != nullptr;
  const int cc = chunk_size(pl, C);
  const size_t af = act_floats_per_chain(m);
  const size_t ws_floats = 2 * (size_t)cc * af + (size_t)cc + 2 * (size_t)cc * P + (size_t)cc;
  int rc = ensure_work(pl, ws_floats * sizeof(T));
  if (rc) return rc;
  T* ws = (T*)pl->d_work;
  T* lik_tmp = ws + 2 * (size_t)cc * af;
  T* prop = lik_tmp + cc;
  T* gprop = prop + (size_t)cc * P;
  T* tprop = gprop + (size_t)cc * P;
  const T sqrt_step = (T)sqrt(step);  // scale = sqrt(step) in double on the host (mala.py:39)
  for (int64_t c0 = 0; c0 < C; c0 += cc) {
    const int n = (int)((C - c0) < cc ? (C - c0) : cc);
    T* th_c = (T*)theta + c0 * P;
    T* g_c = mala ? (T*)grad + c0 * P : nullptr;
    const T* temp_c = temp ? (const T*)temp + c0 : nullptr;
    const T* sv_c = step_vec ? (const T*)step_vec + c0 : nullptr;
    hipLaunchKernelGGL((k_propose<T>), dim3(n), dim3(256), 0, s, (const T*)th_c, (const T*)g_c,
                       z ? (const T*)z + c0 * P : nullptr, (const T*)scale, prop, P, (T)step, sv_c,
                       sqrt_step, seed, iter, chain_offset + (uint64_t)c0);
    if ((rc = eval_chunk<T>(pl, prop, temp_c, n, nullptr, nullptr, tprop, mala ? gprop : nullptr, ws, lik_tmp, s)))
      return rc;
    hipLaunchKernelGGL((k_mh_finish<T>), dim3(n), dim3(256), 0, s, th_c, g_c, (T*)target + c0, (const T*)prop,
                       mala ? (const T*)gprop : nullptr, (const T*)tprop,
                       u ? (const T*)u + c0 : nullptr, P, (T)step, sv_c, sqrt_step, seed, iter,
                       chain_offset + (uint64_t)c0, (unsigned char*)accepted + c0,
                       log_rate ? (T*)log_rate + c0 : nullptr);
  }
  EY_HIP(hipGetLastError());
  return EY_OK;
}
int ey_large_mala_mh(ey_plan* pl, void* theta, void* target, void* grad, const void* z, const void* u, double step,
                     const void* step_vec, const void* scale, const void* temp, int64_t C, uint64_t seed, uint64_t iter,
                     uint64_t chain_offset, void* accepted, void* log_rate, hipStream_t s) {
  return pl->dtype == EY_F32 ? large_mala_mh<float>(pl, theta, target, grad, z, u, step, step_vec, scale, temp, C, seed, iter, chain_offset, accepted, log_rate, s) : large_mala_mh<double>(pl, theta, target, grad, z, u, step, step_vec, scale, temp, C, seed, iter, chain_offset, accepted, log_rate, s);
}

// HMC.leapfrog (hmc.py:100-124) as the reference runs it: L steps, L+1 gradient evaluations, momentum negated at the end
template <class T>
static int large_leapfrog(ey_plan* pl, void* theta, void* p, double step, const void* step_vec, int L, const void* temp,
                      int64_t C, void* target, void* grad, hipStream_t s) {
  const EyModel& m = pl->m;
  const int P = m.P;
  const int cc = chunk_size(pl, C);
  const size_t af = act_floats_per_chain(m);
  const int nblk = leap_blocks(P);
  const size_t ws_floats = 2 * (size_t)cc * af + (size_t)cc + (size_t)cc * nblk;
  int rc = ensure_work(pl, ws_floats * sizeof(T));
  if (rc) return rc;
  T* ws = (T*)pl->d_work;
  T* lik_tmp = ws + 2 * (size_t)cc * af;
  T* qpart = lik_tmp + cc;
  const T* mu = (const T*)m.mu;
  const T* iv = (const T*)m.inv_var;
  for (int64_t c0 = 0; c0 < C; c0 += cc) {
    const int n = (int)((C - c0) < cc ? (C - c0) : cc);
    T* th_c = (T*)theta + c0 * P;
    T* p_c = (T*)p + c0 * P;
    T* g_c = (T*)grad + c0 * P;
    T* t_c = (T*)target + c0;
    const T* temp_c = temp ? (const T*)temp + c0 : nullptr;
    const T* sv_c = step_vec ? (const T*)step_vec + c0 : nullptr;
    const dim3 grid(nblk, n);
    if ((rc = eval_chunk<T>(pl, th_c, temp_c, n, nullptr, nullptr, t_c, g_c, ws, lik_tmp, s))) return rc;  // :104
    hipLaunchKernelGGL((k_leap<T>), grid, dim3(256), 0, s, th_c, p_c, (const T*)g_c, P, (T)step, sv_c, T(0.5), T(1.0),
                       mu, iv, qpart);
    for (int k = 1; k <= L; ++k) {
      if ((rc = eval_chunk<T>(pl, th_c, temp_c, n, nullptr, nullptr, t_c, g_c, ws, lik_tmp, s, qpart))) return rc;
      hipLaunchKernelGGL((k_leap<T>), grid, dim3(256), 0, s, th_c, p_c, (const T*)g_c, P, (T)step, sv_c,
                         k < L ? T(1.0) : T(0.5), k < L ? T(1.0) : T(0.0), mu, iv, qpart);
    }
    const long tot = (long)n * P;
    hipLaunchKernelGGL((k_negate<T>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, p_c, tot);  // :122
  }
  EY_HIP(hipGetLastError());
  return EY_OK;
}
int ey_large_leapfrog(ey_plan* pl, void* theta, void* p, double step, const void* step_vec, int L, const void* temp,
                      int64_t C, void* target, void* grad, hipStream_t s) {
  return pl->dtype == EY_F32 ? large_leapfrog<float>(pl, theta, p, step, step_vec, L, temp, C, target, grad, s) : large_leapfrog<double>(pl, theta, p, step, step_vec, L, temp, C, target, grad, s);
}

// the N terms of the log-likelihood sum of every chain (ey_log_lik_rows): forward products and the loss kernel only
template <class T>
static int large_log_lik_rows(ey_plan* pl, const void* theta, const void* temp, int64_t C, void* rows, hipStream_t s) {
  const EyModel& m = pl->m;
  const int cc = chunk_size(pl, C);
  const size_t af = act_floats_per_chain(m);
  const size_t ws_floats = 2 * (size_t)cc * af + 2 * (size_t)cc;
  int rc = ensure_work(pl, ws_floats * sizeof(T));
  if (rc) return rc;
  T* ws = (T*)pl->d_work;
  T* lik_tmp = ws + 2 * (size_t)cc * af;
  for (int64_t c0 = 0; c0 < C; c0 += cc) {
    const int n = (int)((C - c0) < cc ? (C - c0) : cc);
    rc = eval_chunk<T>(pl, (const T*)theta + c0 * m.P, temp ? (const T*)temp + c0 : nullptr, n, nullptr, nullptr,
                    lik_tmp + cc, nullptr, ws, lik_tmp, s, nullptr, (T*)rows + c0 * m.N);
    if (rc) return rc;
  }
  return EY_OK;
}
int ey_large_log_lik_rows(ey_plan* pl, const void* theta, const void* temp, int64_t C, void* rows, hipStream_t s) {
  return pl->dtype == EY_F32 ? large_log_lik_rows<float>(pl, theta, temp, C, rows, s) : large_log_lik_rows<double>(pl, theta, temp, C, rows, s);
}

// Test / measurement entry (not part of the sampler surface): C[b] = act(A[b] B[b] + bias[b]) through the same
// dispatcher the evaluation uses.  Strides in elements; bias may be null; act is an EY_ACT_* code.
extern "C" int ey_debug_bgemm(const float* A, const float* B, float* C, int M, int N, int K, long sAm, long sAk, long sBk,
                              long sBn, long sCm, long sCn, long bA, long bB, long bC, const float* bias, long bBias,
                              int act, int batch, void* stream) {
  EyVariantScope vs(ey_default_variant(), ey_default_products());
  BG g = {};
  g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K;
  g.sAm = sAm; g.sAk = sAk; g.sBk = sBk; g.sBn = sBn; g.sCm = sCm; g.sCn = sCn;
  g.bA = bA; g.bB = bB; g.bC = bC; g.bias = bias; g.bBias = bBias; g.act = act;
  return bgemm(g, batch, (hipStream_t)stream);
}
