// Fused trajectory kernels on 16x16x4 matrix-core tiles for three-layer MLPs d0-H-H-dK with H in {16, 32, 64},
// d0 <= 8, dK <= 4, in f32 (v_mfma_f32_16x16x4_f32) and f64 (v_mfma_f64_16x16x4_f64): the models the reference
// builds besides the headline 4-32-32-3 (eeyore/models/mlp.py:9-19,37-50 takes any dims; its own tests are
// BCE nets, tests/test_binary_classif_mlp2321_log_lik.py) and the reference's default dtype (eeyore/models/model.py:7).
// CE-sum on logits or BCE-sum on a sigmoid output (eeyore/constants/constants.py:15-18), hidden activations sigmoid /
// tanh / relu, every layer with a bias.  Same structure as ey_mfma32.hip: one wavefront per chain in a persistent
// workgroup per CU, the whole draw inside one launch, theta, momentum and gradient in registers.
//
// Reference semantics restated (paths relative to papamarkou/eeyore): MLP.forward eeyore/models/mlp.py:45-50,
// losses eeyore/constants/constants.py:15-18 and eeyore/stats/loss.py:1-11 (naive BCE logs kept), log_target
// eeyore/models/bayesian_model.py:30-56, gradient eeyore/models/log_target_model.py:15-23 (autograd there),
// HMC.leapfrog / draw eeyore/samplers/hmc.py:100-156, MALA.draw eeyore/samplers/mala.py:46-82, MetropolisHastings.draw
// eeyore/samplers/metropolis_hastings.py:41-73.
//
// Tiles.  Lane l = (c = l & 15, g = l >> 4).  One 16x16x4 product D += A B takes A[i = c][k = g] and B[k = g][j = c]
// from lane (c, g) and leaves D[i = fi(g, r)][j = c] in register r, where the two dtypes differ (measured with
// tools/mfma_probe.hip):  fi(g, r) = 4g + r for v_mfma_f32_16x16x4_f32,  4r + g for v_mfma_f64_16x16x4_f64.
//   "T tile" m of Q[feature][row]: register r of lane (c, g) = Q[16m + fi(g, r)][row c]    (an accumulator as it stands)
//   "U tile" n of Q             : register r of lane (c, g) = Q[16n + c][row fi(g, r)]     (read back transposed from LDS)
// Every layer is computed transposed (H_l^T = W_l H_{l-1}^T), so a T tile is directly the B operand of the next layer's
// product (k-step (m, r) gives k-slot g the feature 16m + fi(g, r)) and the A operand of the untransposed
// dH0 = delta1 W1, whose accumulator is then a U tile; the products that contract over rows (dW_l) take U tiles, one
// LDS round trip each: a T tile is stored with its row c at column perm(c) of the buffer (perm(4g + r) = fi(g, r), an
// involution), so that the four rows fi(g, 0..3) of a U tile are one aligned vector read.  Skinny dimensions (d0, dK)
// are padded to one 16-wide tile with zeros in the staged operand images.
//
// Canonical registers of theta / gradient (the D layouts of the weight-gradient products), lane (c, g), f = fi(g, r):
//   w1[(mo MT + n) 4 + r] = W1[16mo + f][16n + c]      w0[4m + r] = W0[16m + f][c]   (c < d0)
//   w2[4n + r] = W2[f][16n + c]   (f < dK)             b1[mo] = b1[16mo + c], b0[m] = b0[16m + c]   b2[o] uniform
#include <algorithm>

#include "ey_common.h"

enum { F16_HMC = 0, F16_GRAD = 1, F16_LEAPFROG = 2, F16_MALA = 3, F16_MH = 4 };
// as a template argument only: the HMC draw with its run-time options compiled out -- every parameter under the same prior, no
// temperature, no step-size tuner attached (what `HMC.run` on a plain posterior issues, and the bench)
enum { F16_HMC_PLAIN = 8 };
#define F16_TS 20  // row stride of the transpose buffers: 16 rows + 4 (keeps the 4-element reads aligned)

template <typename T>
struct F16Args {
  int d0, dK, act0, act1, lik, P;
  int h1, h2;  // true widths of the two hidden layers (<= H, the template's tile grid; the rest is zero padding)
  int two;     // a model with ONE hidden layer (d0-h-dK): the middle layer is skipped, its slots hold no parameter
  int iW0, iB0, iW1, iB1, iW2, iB2;  // offsets of the layers in theta (weights row-major, then the bias, per layer)
  const T* xpack;                    // [ntiles][xt] operand-order data image (k_f16_pack)
  int ntiles, ks0, xt;
  const T* mu;
  const T* inv_var;
  int prior_uniform;  // every parameter has the same (mu0, iv0): no per-parameter prior loads (two global loads per
  T mu0, iv0;         // register slot and evaluation otherwise: 12-69 % of a step, tools/ablate_fused16.sh)
  T prior_const;
  int64_t C;
  T* theta;
  T* target;
  T* grad;
  const T* p0;   // HMC: momentum [C,P]; MALA / MH: standard normals [C,P]; null => Philox
  T* pio;        // LEAPFROG: momentum in/out
  const T* u;
  T step, sqrt_step;
  const T* scale;  // MH proposal scale [P]
  const T* step_vec;
  int L;
  const T* temp;
  uint64_t seed, iter, chain_offset;
  int recompute, mode;
  unsigned char* accepted;
  T *rate, *hcur, *hprop;
  int n_iters;
  T* rec_samples;
  T* rec_targets;
  unsigned char* rec_accepted;
  int* accept_count;
  // attached per-chain dual averaging (ey_plan_attach_da)
  double* da_state;
  const double* da_tab;
  T* da_step;
  int da_n, da_final_it, da_has_eub;
  double da_d, da_logeub;
};

// F16_ABLATE (diagnostic builds only, tools/ablate_fused16.sh): 1 = no transcendental in the activations, 2 = the
// skinny products (logits, dH1, dW2, dW0) replaced by one add each, 4 = no global loads of the data tile, 8 = no
// transpose stores.  Results are wrong in such a build; only its timing is read.
// EY_F16_PART: the family is built as two translation units.  0 (this file as it stands) = the host side and every
// instantiation except the one-wave-per-SIMD ones named next; 1 (ey_fused16_d32.hip, which includes this file) =
// k_fused16<double, 32, 4, *> and the HMC kernels of the f32 H = 64 shapes, k_fused16<float, 64, 4, 2 | 3, F16_HMC>, with
// their launchers only; 2 (ey_fused16_plain.hip) = the plain HMC kernels (F16_HMC_PLAIN) of the other shapes, built beside the
// main unit so that they cost the build no time.  The split exists for one compiler flag: at one wave per SIMD (512 registers) the compiler selects the
// MFMAs in their AGPR form and then keeps the loop-carried accumulators in architectural registers all the same, copying
// them in and out around every product (a third of the vector instructions of the f64 tile loop); -mllvm
// -amdgpu-mfma-vgpr-form on that unit removes the copies (+7 % on the f64 headline model, same bits), and cannot be given
// to the whole file because the same compiler crashes with it on k_fused16<float, 64, 4, 2> with the mode a run-time
// argument.
#ifndef EY_F16_PART
#define EY_F16_PART 0
#endif
#ifndef EY_F16_W16
#define EY_F16_W16 16  // waves per CU of the f32 H = 16 instantiations (see f16_launch)
#endif
#ifndef EY_F16_PLAIN_UNIT
#define EY_F16_PLAIN_UNIT 1  // 0: A/B builds without ey_fused16_plain.hip's kernels
#endif
#ifndef EY_F16_HMC_OWN
#define EY_F16_HMC_OWN 4  // the instantiations of up to this many waves per CU get an HMC kernel of their own (f16_launch_t)
#endif
#ifndef EY_F16_TANH_EM1
#define EY_F16_TANH_EM1 1  // the f64 tanh on expm1 (0: round 3's form, kept for A/B)
#endif
#ifndef F16_ABLATE
#define F16_ABLATE 0
#endif

template <typename T>
struct V4 {
  typedef T type __attribute__((ext_vector_type(4)));
};
template <typename T>
using v4 = typename V4<T>::type;

template <typename T>
__device__ __forceinline__ v4<T> mfma16(T a, T b, v4<T> c);
// v_mfma_f64_4x4x4_4b_f64 (measured, tools/mfma44_probe.hip: 15 cycles against 58 for 16x16x4): four blocks b of
// D_b[i][j] += sum_k A_b[i][k] B_b[k][j]; A_b[i][k] is lane 16k + 4b + i, B_b[k][j] lane 16k + 4b + j, D_b[i][j] lane
// 16i + 4b + j, one register each.  In the coordinates of the 16x16x4 layouts -- lane (c, g) = 16g + c -- the B operand is
// a T or U tile register as it stands (k-slot g, column c = 4b + j), the A operand a tile whose column index only counts
// modulo four (row i = c & 3 of a four-row matrix, repeated for every block), and D is register 0 of a D tile whose rows
// 0 .. 3 are the lane groups: exactly what the products with at most four outputs need (f64: output o lives in lane group o).
__device__ __forceinline__ double mfma44(double a, double b, double c) {
  if (F16_ABLATE & 2) return c + a * b;
  return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}
template <typename T>
__device__ __forceinline__ v4<T> mfma16s(T a, T b, v4<T> c) {  // the skinny products
  if (F16_ABLATE & 2) { c[0] += a * b; return c; }
  return mfma16<T>(a, b, c);
}
template <typename T>
__device__ __forceinline__ v4<T> mfma16(T a, T b, v4<T> c) {
  if constexpr (sizeof(T) == 8) return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// D-register layout of the two 16x16x4 instructions (see the header)
template <typename T>
struct Lay {
  static __host__ __device__ __forceinline__ int fi(int g, int r) { return sizeof(T) == 8 ? 4 * r + g : 4 * g + r; }
  static __host__ __device__ __forceinline__ int perm(int j) { return fi(j >> 2, j & 3); }  // position 4g + r <-> fi(g, r)
};

template <typename T>
struct Nm;
template <>
struct Nm<float> {
  static __device__ __forceinline__ float exp(float v) { return __expf(v); }  // v_exp_f32 / v_log_f32, as ey_mfma32.hip
  static __device__ __forceinline__ float exp_fast(float v) { return __expf(v); }
  static __device__ __forceinline__ float log(float v) { return __logf(v); }
  static __device__ __forceinline__ float tanh(float v) { return tanhf(v); }
  static __device__ __forceinline__ float sqrt(float v) { return sqrtf(v); }
};
// exp in f64 for the activations and the softmax: n = rint(x log2 e), r = x - n ln 2 in two pieces (|r| <= 0.3466), the
// Taylor polynomial of degree 13 (remainder 4e-18 relative), v_ldexp_f64 (which overflows to inf and underflows through
// the denormals to 0 as exp does; NaN passes through): 24 instructions where the library's takes about 45, accurate to a
// few ulp -- the reference's f64 values to 1e-10 need 1e5 times less.
__device__ __forceinline__ double f16_exp_f64(double x) {
  // the argument clamped to where exp leaves the doubles anyway (exp(-746) = 0, exp(710) = inf through v_ldexp_f64): -inf /
  // +inf and products that overflow give the library's 0 / inf instead of inf - inf = NaN in the reduction; NaN passes
  // through (v_max / v_min return the other operand for a quiet NaN, so the select keeps it explicit)
  const double xc = x != x ? x : __builtin_fmax(__builtin_fmin(x, 710.0), -746.0);
  x = xc;
  const double n = __builtin_rint(x * 1.4426950408889634);
  double r = __builtin_fma(-n, 6.93147180369123816490e-01, x);
  r = __builtin_fma(-n, 1.90821492927058770002e-10, r);
  double p = 1.0 / 6227020800.0;
  p = __builtin_fma(p, r, 1.0 / 479001600.0);
  p = __builtin_fma(p, r, 1.0 / 39916800.0);
  p = __builtin_fma(p, r, 1.0 / 3628800.0);
  p = __builtin_fma(p, r, 1.0 / 362880.0);
  p = __builtin_fma(p, r, 1.0 / 40320.0);
  p = __builtin_fma(p, r, 1.0 / 5040.0);
  p = __builtin_fma(p, r, 1.0 / 720.0);
  p = __builtin_fma(p, r, 1.0 / 120.0);
  p = __builtin_fma(p, r, 1.0 / 24.0);
  p = __builtin_fma(p, r, 1.0 / 6.0);
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return __builtin_amdgcn_ldexp(p, (int)n);
}
template <>
struct Nm<double> {
  static __device__ __forceinline__ double exp(double v) { return ::exp(v); }
  static __device__ __forceinline__ double exp_fast(double v) { return f16_exp_f64(v); }
  static __device__ __forceinline__ double log(double v) { return ::log(v); }
  static __device__ __forceinline__ double tanh(double v) { return ::tanh(v); }
  static __device__ __forceinline__ double sqrt(double v) { return ::sqrt(v); }
};
// 1 / d for d in [1, inf] in f64: v_rcp_f64 and two Newton steps (~1 ulp) where the IEEE division is thirteen
// instructions.  d is capped at 1e300 for the iteration (inf * 0 would be NaN) and d = inf returns exactly 0: a sigmoid
// whose exp overflowed is exactly 0, and the reference's naive BCE logs then give -inf / NaN, which rejects
// (eeyore/stats/loss.py:2).  (Between 1e300 and the overflow, logits of 691 .. 709, the result is 1e-300.)
template <bool EXACT_ZERO = true>
__device__ __forceinline__ double f16_recip_ge1(double d) {
  const double dc = fmin(d, 1e300);
  double y = __builtin_amdgcn_rcp(dc);
  y = __builtin_fma(y, __builtin_fma(-dc, y, 1.0), y);
  y = __builtin_fma(y, __builtin_fma(-dc, y, 1.0), y);
  // (hidden units and the softmax do without the select: 1e-300 for an exact 0 reaches nothing at f64 precision)
  return EXACT_ZERO && d > 1.7e308 ? 0.0 : y;
}
// sigmoid and tanh.  f64: the library exp / tanh, as the generic kernels (1e-10 parity with the reference's fp64).  f32:
// one v_exp_f32 and one v_rcp_f32 per element, as ey_mfma32.hip (an IEEE division is ten vector instructions, and f32
// MFMA shares the vector ALUs with them); both are accurate to ~1 ulp, far inside the stated 2e-4.
template <typename T, bool OUTPUT = true>  // OUTPUT: a sigmoid whose saturation to exactly 0 matters (the BCE head)
__device__ __forceinline__ T f16_sigmoid(T g) {
  if (F16_ABLATE & 1) return T(0.25) * g + T(0.5);
  if constexpr (sizeof(T) == 4) return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * g));
  else return f16_recip_ge1<OUTPUT>(T(1) + Nm<T>::exp_fast(-g));
}
template <typename T>
__device__ __forceinline__ T f16_tanh(T g) {
  if constexpr (sizeof(T) == 4) return 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * g)) - 1.0f;
  else {
#if EY_F16_TANH_EM1
    // tanh |g| = -m / (2 + m) with m = expm1(-2 |g|), so that nothing cancels near 0: exp(t) = 2^n (1 + r q(r)) with
    // n = rint(t log2 e), |r| <= 0.3466 and q the Taylor polynomial of (e^r - 1) / r to r^12 (remainder 4e-18); for n = 0
    // (|g| < 0.173) m = r q exactly as computed, otherwise m = 2^n (1 + r q) - 1 with 2^n <= 1/2.  2 + m lies in (1, 2], so
    // the reciprocal is v_rcp_f64 and two Newton steps without a range guard.  36 instructions per element (the form it
    // replaces -- the exp, 2 / (1 + e) - 1, and a separate odd polynomial below |g| = 0.125 -- took 48), at most 3 ulp from
    // the library's tanh over [-45, 45] and down to 2^-60 (checked on the host with the same fma sequence); a NaN stays a
    // NaN (the clamp is a compare-and-select, which keeps it; v_max would return the bound), +-inf give +-1.
    T t = T(-2) * __builtin_fabs(g);
    t = t < T(-80) ? T(-80) : t;  // exp(-80) = 1.8e-35: tanh is 1 to the last bit long before
    const T n = __builtin_rint(t * T(1.4426950408889634));
    T r = __builtin_fma(-n, T(6.93147180369123816490e-01), t);
    r = __builtin_fma(-n, T(1.90821492927058770002e-10), r);
    T q = T(1.0 / 6227020800.0);
    q = __builtin_fma(q, r, T(1.0 / 479001600.0));
    q = __builtin_fma(q, r, T(1.0 / 39916800.0));
    q = __builtin_fma(q, r, T(1.0 / 3628800.0));
    q = __builtin_fma(q, r, T(1.0 / 362880.0));
    q = __builtin_fma(q, r, T(1.0 / 40320.0));
    q = __builtin_fma(q, r, T(1.0 / 5040.0));
    q = __builtin_fma(q, r, T(1.0 / 720.0));
    q = __builtin_fma(q, r, T(1.0 / 120.0));
    q = __builtin_fma(q, r, T(1.0 / 24.0));
    q = __builtin_fma(q, r, T(1.0 / 6.0));
    q = __builtin_fma(q, r, T(0.5));
    q = __builtin_fma(q, r, T(1.0));
    const T rq = r * q;
    const T e = __builtin_amdgcn_ldexp(__builtin_fma(r, q, T(1)), (int)n);
    const T m = n == T(0) ? rq : e - T(1);
    const T d = T(2) + m;
    T y = __builtin_amdgcn_rcp(d);
    y = __builtin_fma(y, __builtin_fma(-d, y, T(1)), y);
    y = __builtin_fma(y, __builtin_fma(-d, y, T(1)), y);
    return __builtin_copysign(-m * y, g);
#else
    // tanh |g| = 2 / (1 + exp(-2 |g|)) - 1 on the 19-instruction exp (the library's tanh is 139 f64 instructions per
    // element); below |g| = 0.125, where that form cancels, the odd Taylor polynomial to g^15 (remainder 2e-18 relative)
    const T big = T(2) * f16_recip_ge1<false>(T(1) + Nm<T>::exp_fast(-T(2) * __builtin_fabs(g))) - T(1);
    const T g2 = g * g;
    T p = T(-929569.0 / 638512875.0);
    p = __builtin_fma(p, g2, T(21844.0 / 6081075.0));
    p = __builtin_fma(p, g2, T(-1382.0 / 155925.0));
    p = __builtin_fma(p, g2, T(62.0 / 2835.0));
    p = __builtin_fma(p, g2, T(-17.0 / 315.0));
    p = __builtin_fma(p, g2, T(2.0 / 15.0));
    p = __builtin_fma(p, g2, T(-1.0 / 3.0));
    const T small = __builtin_fma(p * g2, g, g);
    return __builtin_fabs(g) < T(0.125) ? small : __builtin_copysign(big, g);
#endif
  }
}
template <typename T>
__device__ __forceinline__ T f16_act(int code, T g) {
  switch (code) {
    case EY_ACT_SIGMOID: return f16_sigmoid<T>(g);
    case EY_ACT_TANH: return f16_tanh<T>(g);
    case EY_ACT_RELU: return g > T(0) ? g : T(0);
    default: return g;
  }
}
template <typename T>
__device__ __forceinline__ T f16_dact(int code, T h) {
  switch (code) {
    case EY_ACT_SIGMOID: return h * (T(1) - h);
    case EY_ACT_TANH: return T(1) - h * h;
    case EY_ACT_RELU: return h > T(0) ? T(1) : T(0);
    default: return T(1);
  }
}
// the activation of whole tiles with the (wave-uniform) switch outside the element loop: one straight-line body per
// activation instead of a three-way select around every element
template <typename T, int MT>
__device__ __forceinline__ void f16_act_tiles(int code, typename V4<T>::type (&h)[MT]) {
  if (code == EY_ACT_SIGMOID) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) h[m][r] = f16_sigmoid<T, false>(h[m][r]);
  } else if (code == EY_ACT_TANH) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) h[m][r] = f16_tanh<T>(h[m][r]);
  } else if (code == EY_ACT_RELU) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) h[m][r] = h[m][r] > T(0) ? h[m][r] : T(0);
  }
}
// d *= act'(h) for whole tiles
template <typename T, int MT>
__device__ __forceinline__ void f16_dact_tiles(int code, typename V4<T>::type (&d)[MT], const typename V4<T>::type (&h)[MT]) {
  if (code == EY_ACT_SIGMOID) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) d[m][r] *= h[m][r] * (T(1) - h[m][r]);
  } else if (code == EY_ACT_TANH) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) d[m][r] *= T(1) - h[m][r] * h[m][r];
  } else if (code == EY_ACT_RELU) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) d[m][r] *= h[m][r] > T(0) ? T(1) : T(0);
  }
}

// wave total in every lane, without LDS traffic: DPP adds inside each 16-lane row (quad swaps, half-mirror, mirror), then
// row_bcast:15 / row_bcast:31 carry the row totals to lane 63, whose value v_readlane hands to all (as wsum of
// ey_mfma32.hip; six __shfl_xor rounds were twelve ds_bpermute in f64)
template <int CTRL, int ROWMASK, typename T>
__device__ __forceinline__ T f16_dpp(T v) {
  if constexpr (sizeof(T) == 4) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xF, false));
  } else {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, ROWMASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROWMASK, 0xF, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
  }
}
template <typename T>
__device__ __forceinline__ T f16_wsum(T v) {
#ifdef F16_WSUM_SHFL  // diagnostic builds: the butterfly through ds_bpermute instead (a different order of additions)
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
#endif
  v += f16_dpp<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v += f16_dpp<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v += f16_dpp<0x141, 0xF>(v);  // row_half_mirror
  v += f16_dpp<0x140, 0xF>(v);  // row_mirror
  v += f16_dpp<0x142, 0xA>(v);  // row_bcast:15
  v += f16_dpp<0x143, 0xC>(v);  // row_bcast:31
  if constexpr (sizeof(T) == 4) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
  } else {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)b, 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
  }
}
// sum over the four lane groups g (lanes c, c + 16, c + 32, c + 48): every lane gets the total of its column c
template <typename T>
__device__ __forceinline__ T f16_gsum(T v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ void f16_fence() {
  // LDS operations of one wave execute in issue order; this only stops the compiler from moving them
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// KSM: k-steps (of four inputs) the first layer's operand image is laid out for -- 2 in the exact-shape instantiations
// (their LDS layout is untouched by the wider forms), 4 in the padded ones
template <int H, int KSM = 2>
struct F16Cfg {
  static constexpr int MT = H / 16;
  static constexpr int NW1 = MT * MT * 4, NW0 = MT * 4, NW2 = MT * 4;
  static constexpr int S_W0 = NW1, S_W2 = S_W0 + NW0, S_B1 = S_W2 + NW2, S_B0 = S_B1 + MT, S_B2 = S_B0 + MT;
  static constexpr int NREG = S_B2 + 4;
  // per-wave LDS carve, in elements
  static constexpr int O_W1A = 0;                       // [(mo MT + m) 4 + r][lane]: A operands of F1
  static constexpr int O_TB0 = H * H;                   // [H][TS] transpose buffer: H1, then delta1
  static constexpr int O_TB1 = O_TB0 + H * F16_TS;      // [H][TS] transpose buffer: H0
  static constexpr int O_W0A = O_TB1 + H * F16_TS;      // [m][s < KSM][lane]: A operands of F0 (up to 4 KSM inputs)
  static constexpr int O_W2A = O_W0A + MT * KSM * 64;   // [m][r][lane]: A operands of the logits
  // (the padded configuration, KSM = 4, also serves 5 .. 16 outputs: four k-steps of dH1 and a whole delta2 tile)
  static constexpr int O_W2T = O_W2A + MT * 4 * 64;     // [m][lane] ([m][s][lane]): A operands of dH1
  static constexpr int O_B0 = O_W2T + MT * 64 * (KSM == 4 ? 4 : 1);  // [H]
  static constexpr int O_B1 = O_B0 + H;                 // [H]
  static constexpr int O_D2 = O_B1 + H;                 // [4][16] delta2[o][row]  ([16][TS] transpose buffer of the delta2 tile)
  // O_JUNK: one element per lane that nothing reads.  A store that a lane must not make (a feature, an output or a bias slot
  // the lane does not own) goes there instead of sitting behind a per-lane branch: inside such a branch only some lanes are
  // active, and a register the allocator spills and reloads there -- theta, the momentum, a gradient: everything is live
  // across these stores -- comes back with the inactive lanes' values lost.
  static constexpr int O_JUNK = O_D2 + (KSM == 4 ? 16 * F16_TS : 64);
  static constexpr int WAVE_ELEMS = O_JUNK + 64;
  // the N(0,1) stream of a draw is staged in [O_W1A, O_W0A) between evaluations: P <= H^2 + 14 H + 4 <= H^2 + 40 H
  static_assert(H * H + 14 * H + 4 <= O_W0A, "the staging area must hold P normals");
};

struct F16Slot {
  int idx;
  bool valid, counts;
};
// canonical element k of lane (c, g): its index in theta, whether this lane holds anything there, and whether this
// lane's copy is the one that enters sums over parameters
template <int H, typename T, int V, typename A>
__device__ __forceinline__ F16Slot f16_slot(int k, const A& a, int c, int g, int lane) {
  typedef F16Cfg<H> K;
  // V & 1 (TWO): one hidden layer, the middle layer's slots hold nothing.  V & 2 (PAD): hidden widths h1, h2 below the
  // tile grid H -- the slots of features beyond them hold no parameter (theta 0, gradient masked, no momentum).  A
  // padded hidden unit is act(0), not 0 for a sigmoid, but every weight that reads it is a padding slot, so it reaches
  // nothing; the weight gradients it produces land in padding slots and are dropped.  Without PAD the widths ARE H and
  // the strides are compile-time constants (the run-time form cost the exact shapes up to 4.5 %).
  constexpr bool TWO = (V & 1) != 0, PAD = (V & 2) != 0;
  const int h1 = PAD ? a.h1 : H, h2 = PAD ? a.h2 : H;
  if (k < K::S_W0) {
    const int r = k & 3, n = (k >> 2) % K::MT, mo = (k >> 2) / K::MT;
    const int out = 16 * mo + Lay<T>::fi(g, r), in = 16 * n + c;
    const bool v = !TWO && (!PAD || (out < h2 && in < h1));
    return {a.iW1 + out * h1 + in, v, v};
  }
  if (k < K::S_W2) {
    const int kk = k - K::S_W0, r = kk & 3, m = kk >> 2;
    const int out = 16 * m + Lay<T>::fi(g, r);
    const bool v = c < a.d0 && (!PAD || out < h1);
    return {a.iW0 + out * a.d0 + c, v, v};
  }
  if (k < K::S_B1) {
    const int kk = k - K::S_W2, r = kk & 3, n = kk >> 2;
    const int in = 16 * n + c;
    const bool v = Lay<T>::fi(g, r) < a.dK && (!PAD || in < h2);
    return {a.iW2 + Lay<T>::fi(g, r) * h2 + in, v, v};
  }
  if (k < K::S_B0) {
    const int f = 16 * (k - K::S_B1) + c;
    const bool v = !TWO && (!PAD || (f < h2 && a.iB1 >= 0));  // (a layer without a bias, mlp.py:40-42: its slots hold nothing)
    return {a.iB1 + f, v, v && g == 0};
  }
  if (k < K::S_B2) {
    const int f = 16 * (k - K::S_B0) + c;
    const bool v = !PAD || (f < h1 && a.iB0 >= 0);
    return {a.iB0 + f, v, v && g == 0};
  }
  const int o = k - K::S_B2;
  if (PAD && a.dK > 4) {  // more than four outputs: slot r holds b2[fi(g, r)], the D layout of the logits tile
    const int cls = Lay<T>::fi(g, o);
    const bool v = cls < a.dK && a.iB2 >= 0;
    return {a.iB2 + cls, v, v && c == 0};
  }
  const bool v = o < a.dK && (!PAD || a.iB2 >= 0);
  return {a.iB2 + o, v, lane == 0 && v};
}
#define F16_EACH(k) _Pragma("unroll") for (int k = 0; k < K::NREG; ++k)

// stage the operand images of the position `th` in this wave's LDS region (the zero padding of the images was
// written once at kernel start and is never overwritten).  An element W[out][in] this lane holds goes where the lane
// that feeds it to the product will read it: A operand lane (i & 15, k-slot), one image row of 64 per k-step.
template <typename T, int H, int KSM, typename A>
__device__ __forceinline__ void f16_write_images(T* lw, const T (&th)[F16Cfg<H>::NREG], const A& a, int c, int g) {
  typedef F16Cfg<H, KSM> K;
  typedef Lay<T> L;
  // the k-step register rk and k-slot gk at which input feature (16n +) c enters a product whose B operand is a T tile
  const int pc = L::perm(c), rk = pc & 3, gk = pc >> 2;
#pragma unroll
  for (int k = 0; k < K::NW1; ++k) {  // W1[16mo + fi(g, r)][16n + c]: A operand of F1, k-step (n, rk)
    const int r = k & 3, n = (k >> 2) % K::MT, mo = (k >> 2) / K::MT;
    lw[K::O_W1A + ((mo * K::MT + n) * 4 + rk) * 64 + L::fi(g, r) + 16 * gk] = th[k];
  }
  const int junk = K::O_JUNK + (int)(threadIdx.x & 63);
  {
    const bool own = c < a.d0;
#pragma unroll
    for (int k = 0; k < K::NW0; ++k) {  // W0[16m + fi(g, r)][c]: A operand of F0, k-step c >> 2, k-slot c & 3 (as k_f16_pack lays x out)
      const int r = k & 3, m = k >> 2;
      lw[own ? K::O_W0A + (m * KSM + (c >> 2)) * 64 + L::fi(g, r) + 16 * (c & 3) : junk] = th[K::S_W0 + k];
    }
  }
#pragma unroll
  for (int k = 0; k < K::NW2; ++k) {  // W2[o = fi(g, r)][16n + c]
    const int r = k & 3, n = k >> 2, o = L::fi(g, r);
    const bool own = o < a.dK;
    if (sizeof(T) == 8 && !(KSM == 4 && a.dK > 4)) {
      // f64, at most four outputs: the logits are a 4x4x4 product (mfma44): row o of W2 in every block of four lanes
#pragma unroll
      for (int b4 = 0; b4 < 4; ++b4) lw[own ? K::O_W2A + (n * 4 + rk) * 64 + 4 * b4 + o + 16 * gk : junk] = th[K::S_W2 + k];
    } else {
      lw[own ? K::O_W2A + (n * 4 + rk) * 64 + o + 16 * gk : junk] = th[K::S_W2 + k];  // logits: A lane (o, gk), k-step (n, rk)
    }
    if (KSM == 4 && a.dK > 4)  // dH1 over four k-steps: k-step s, k-slot g' <-> output fi(g', s), which is (r, g) here
      lw[own ? K::O_W2T + (n * 4 + r) * 64 + c + 16 * g : junk] = th[K::S_W2 + k];
    else
      lw[own ? K::O_W2T + n * 64 + c + 16 * o : junk] = th[K::S_W2 + k];            // dH1: A lane (f & 15, k-slot o)
  }
  {  // bias images in the order a vector read at 4g hands out the features fi(g, 0..3): lane group 0 writes them
    const bool own = g == 0;
#pragma unroll
    for (int m = 0; m < K::MT; ++m) {
      lw[own ? K::O_B1 + 16 * m + pc : junk] = th[K::S_B1 + m];
      lw[own ? K::O_B0 + 16 * m + pc : junk] = th[K::S_B0 + m];
    }
  }
  f16_fence();
}

template <typename T>
__device__ __forceinline__ v4<T> f16_ld4(const T* p) {
  return *reinterpret_cast<const v4<T>*>(p);
}

// log-target and gradient of the position whose images are staged in lw; returns the (tempered) log-target.
// GRAD (wave-uniform) = false: the value only (random-walk MH).  Inlined at its three call sites: theta and the gradient
// are register arrays, which a call would force into scratch memory.
template <typename T, int H, int V, bool UPRIOR = false, typename A>
__device__ __forceinline__ T f16_eval(const A& a, T* lw, const T (&th)[F16Cfg<H>::NREG],
                                      T (&gr)[F16Cfg<H>::NREG], const bool GRAD, bool has_temp, T temp, int c, int g,
                                      int lane, T* lik_out = nullptr, T* prior_out = nullptr, const bool need_value = true) {
  typedef F16Cfg<H, (V & 2) ? 4 : 2> K;
  typedef Lay<T> L;
  constexpr int MT = K::MT;
  const int pc = L::perm(c);  // this lane's row c sits at column pc of the transpose buffers
  v4<T> dW1[MT * MT], dW0[MT], dW2[MT];
  T db1[MT], db0[MT], db2[4] = {T(0), T(0), T(0), T(0)}, lik = T(0);
#pragma unroll
  for (int i = 0; i < MT * MT; ++i) dW1[i] = v4<T>{0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    dW0[i] = v4<T>{0, 0, 0, 0};
    dW2[i] = v4<T>{0, 0, 0, 0};
    db1[i] = db0[i] = T(0);
  }
  const T b2v[4] = {th[K::S_B2], th[K::S_B2 + 1], th[K::S_B2 + 2], th[K::S_B2 + 3]};
  const int off_xu = a.ks0 * 64, off_lab = off_xu + 256, off_y = off_lab + 64;
#pragma unroll 1
  for (int t = 0; t < a.ntiles; ++t) {
    const T* xt = a.xpack + (size_t)t * a.xt;
    // k-steps of F0 (four inputs each): two in the exact-shape instantiations (d0 <= 8), four in the padded ones, which
    // also serve 9 .. 16 inputs
    constexpr int KS = (V & 2) ? 4 : 2;
    T xb[KS];
#pragma unroll
    for (int s4 = 0; s4 < KS; ++s4) xb[s4] = (F16_ABLATE & 4) ? T(0.01 * lane) : (s4 < a.ks0 ? xt[s4 * 64 + lane] : T(0));
    const int lab = (F16_ABLATE & 4) ? (c % 3) : (int)xt[off_lab + c];  // -1 marks a padding row
    const bool valid = lab >= 0;
    // ---- F0: H0^T = act0(W0 X^T + b0)                                  (mlp.py:45-50)
    v4<T> H0[MT], H1[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      v4<T> acc = f16_ld4(lw + K::O_B0 + 16 * m + 4 * g);
      acc = mfma16<T>(lw[K::O_W0A + (m * KS) * 64 + lane], xb[0], acc);
#pragma unroll
      for (int s4 = 1; s4 < KS; ++s4)
        if (s4 < a.ks0) acc = mfma16<T>(lw[K::O_W0A + (m * KS + s4) * 64 + lane], xb[s4], acc);
      H0[m] = acc;
    }
    f16_act_tiles<T, MT>(a.act0, H0);
    if (GRAD && !(F16_ABLATE & 8)) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) lw[K::O_TB1 + (16 * m + L::fi(g, r)) * F16_TS + pc] = H0[m][r];
    }
    // ---- F1: H1^T = act1(W1 H0^T + b1)   (one hidden layer: H1 is H0, and a.act1 = a.act0 for the derivative below)
    if constexpr ((V & 1) != 0) {
#pragma unroll
      for (int mo = 0; mo < MT; ++mo) H1[mo] = H0[mo];
    } else {
      // the MT output blocks are independent accumulation chains: stepped together (block innermost), so that at one wave
      // per SIMD an MFMA does not wait for the result of the one before it
#pragma unroll
      for (int mo = 0; mo < MT; ++mo) H1[mo] = f16_ld4(lw + K::O_B1 + 16 * mo + 4 * g);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int mo = 0; mo < MT; ++mo)
            H1[mo] = mfma16<T>(lw[K::O_W1A + ((mo * MT + m) * 4 + r) * 64 + lane], H0[m][r], H1[mo]);
      f16_act_tiles<T, MT>(a.act1, H1);
    }
    if (GRAD && !(F16_ABLATE & 8)) {
#pragma unroll
      for (int mo = 0; mo < MT; ++mo)
#pragma unroll
        for (int r = 0; r < 4; ++r) lw[K::O_TB0 + (16 * mo + L::fi(g, r)) * F16_TS + pc] = H1[mo][r];
    }
    // ---- output layer: logits[o][row c] = W2 H1^T + b2 comes out in register r of lane group g with fi(g, r) = o;
    // every lane then fetches the dK logits of its row c, so all four groups carry the same softmax
    // WIDE (the padded instantiations, 5 .. 16 outputs, CE): the whole 16 x 16 logits tile is live -- register r of lane
    // (c, g) is output fi(g, r) of row c -- and the softmax, delta2 and the products that consume it work on the tile
    const bool WIDE = (V & 2) != 0 && a.dK > 4;  // wave-uniform
    v4<T> lacc;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = L::fi(g, r);  // < 4 only for one (g, r) pair per output: a select, not an indexed read
      lacc[r] = WIDE ? th[K::S_B2 + r]
                     : (o >= a.dK ? T(0) : (o == 0 ? b2v[0] : (o == 1 ? b2v[1] : (o == 2 ? b2v[2] : b2v[3]))));
    }
    // Q44 (f64, at most four outputs): logits and dW2 as 4x4x4 products on one register (see mfma44)
    const bool Q44 = sizeof(T) == 8 && !WIDE;
    if constexpr (sizeof(T) == 8) {
      if (Q44) {
        double lq = lacc[0];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) lq = mfma44(lw[K::O_W2A + (m * 4 + r) * 64 + lane], H1[m][r], lq);
        lacc[0] = lq;
      }
    }
    if (!Q44) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) lacc = mfma16s<T>(lw[K::O_W2A + (m * 4 + r) * 64 + lane], H1[m][r], lacc);
    }
    T lg[4];
    if (!WIDE) {
#pragma unroll
      for (int o = 0; o < 4; ++o) {  // o = fi(go, ro): f32 (0, o), f64 (o, 0)
        const int go = sizeof(T) == 8 ? o : 0, ro = sizeof(T) == 8 ? 0 : o;
        lg[o] = __shfl(lacc[ro], c + 16 * go, 64);
      }
    }
    // ---- log-likelihood and the output delta                                (constants.py:15-18, loss.py:1-11)
    T d2[4] = {T(0), T(0), T(0), T(0)};  // WIDE: the delta2 tile (register r = output fi(g, r)), in every lane group
    const bool mine = valid && g == 0;
    if (WIDE) {
      // row maximum, sum of exponentials and the label's logit: over the lane's four outputs, then over the four lane
      // groups of the row (two exchange rounds each); one exp per live (output, row) pair, no redundancy
      T mx = T(-1e300);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (L::fi(g, r) < a.dK) mx = fmax(mx, lacc[r]);
      mx = fmax(mx, __shfl_xor(mx, 16, 64));
      mx = fmax(mx, __shfl_xor(mx, 32, 64));
      T e[4], ssum = T(0), llab = T(0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = L::fi(g, r);
        e[r] = o < a.dK ? Nm<T>::exp_fast(lacc[r] - mx) : T(0);
        ssum += e[r];
        if (o == lab) llab = lacc[r];
      }
      ssum = f16_gsum(ssum);
      if (need_value) {
        llab = f16_gsum(llab);
        if (mine) lik += llab - (mx + Nm<T>::log(ssum));
      }
      T inv;
      if constexpr (sizeof(T) == 8) inv = f16_recip_ge1<false>(ssum);
      else inv = T(1) / ssum;
#pragma unroll
      for (int r = 0; r < 4; ++r) d2[r] = valid ? ((L::fi(g, r) == lab ? T(1) : T(0)) - e[r] * inv) : T(0);
    } else if (a.lik == EY_LIK_CE_SUM) {
      T mx = lg[0];
#pragma unroll
      for (int o = 1; o < 4; ++o)
        if (o < a.dK) mx = fmax(mx, lg[o]);
      T e[4], ssum = T(0), llab = T(0);
      if constexpr (sizeof(T) == 8) {
        // the library exp is ~50 f64 instructions: lane group g takes output g of its row, the four groups exchange
        const T mine_lg = g == 0 ? lg[0] : (g == 1 ? lg[1] : (g == 2 ? lg[2] : lg[3]));
        const T e_own = g < a.dK ? Nm<T>::exp_fast(mine_lg - mx) : T(0);
#pragma unroll
        for (int o = 0; o < 4; ++o) e[o] = __shfl(e_own, c + 16 * o, 64);
      } else {
#pragma unroll
        for (int o = 0; o < 4; ++o) e[o] = o < a.dK ? Nm<T>::exp(lg[o] - mx) : T(0);
      }
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        ssum += e[o];  // (zero beyond dK)
        if (o == lab) llab = lg[o];
      }
      if (need_value && mine) lik += llab - (mx + Nm<T>::log(ssum));
      T inv;
      if constexpr (sizeof(T) == 8) inv = f16_recip_ge1<false>(ssum);  // the largest term is exp(0)
      else inv = T(1) / ssum;
#pragma unroll
      for (int o = 0; o < 4; ++o)
        if (o < a.dK && mine) d2[o] = (o == lab ? T(1) : T(0)) - e[o] * inv;
    } else {
#pragma unroll
      for (int o = 0; o < 4; ++o)
        if (o < a.dK) {
          const T pr = f16_act<T>(EY_ACT_SIGMOID, lg[o]);
          const T yy = xt[off_y + o * 16 + c];
          // naive logs exactly as eeyore/stats/loss.py:2 (NaN once a sigmoid saturates)
          if (need_value && mine) lik += Nm<T>::log(pr) * yy + Nm<T>::log(T(1) - pr) * (T(1) - yy);
          if (mine) {
            d2[o] = (yy / pr - (T(1) - yy) / (T(1) - pr)) * f16_dact<T>(EY_ACT_SIGMOID, pr);
          }
        }
    }
    if (!GRAD) continue;
    v4<T> D1[MT];
    if (WIDE) {
      // db2 summed in the tile layout (reduced over the rows once per evaluation); the tile itself to its transpose buffer
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        db2[r] += d2[r];
        lw[K::O_D2 + L::fi(g, r) * F16_TS + pc] = d2[r];
      }
      f16_fence();
      // ---- dH1^T = W2^T delta2^T over four k-steps (k-step s, k-slot g <-> output fi(g, s): B = register s of the tile)
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        v4<T> acc = {0, 0, 0, 0};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
          if (sizeof(T) == 4 || 4 * s4 < a.dK)  // (f64: k-step s holds outputs 4s .. 4s+3)
            acc = mfma16s<T>(lw[K::O_W2T + (m * 4 + s4) * 64 + lane], d2[s4], acc);
        D1[m] = acc;
      }
      f16_dact_tiles<T, MT>(a.act1, D1, H1);
      // ---- dW2[o][f] += sum_n delta2[n][o] H1[n][f]: lane c of the U tile is output c (rows of outputs >= dK are zero)
      const v4<T> d2u = f16_ld4(lw + K::O_D2 + c * F16_TS + 4 * g);
#pragma unroll
      for (int n = 0; n < MT; ++n) {
        const v4<T> h1u = f16_ld4(lw + K::O_TB0 + (16 * n + c) * F16_TS + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) dW2[n] = mfma16s<T>(d2u[r], h1u[r], dW2[n]);
      }
    } else {
    {  // (lane group 0 holds the row's delta2: the others add zeros and store to their junk slots -- no per-lane branch)
      const bool own = g == 0;
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        db2[o] += own ? d2[o] : T(0);
        lw[own ? K::O_D2 + o * 16 + pc : K::O_JUNK + lane] = d2[o];
      }
    }
    f16_fence();
    // ---- dH1^T = W2^T delta2^T, delta1 = dH1 * act1'(H1)
    T d2all[4];  // the row's delta2 in every lane group (d2 itself is masked to g = 0 for the sums)
#pragma unroll
    for (int o = 0; o < 4; ++o) d2all[o] = __shfl(d2[o], c, 64);
    const T d2b = g == 0 ? d2all[0] : (g == 1 ? d2all[1] : (g == 2 ? d2all[2] : d2all[3]));  // delta2[row c][o = g]
#pragma unroll
    for (int m = 0; m < MT; ++m) D1[m] = mfma16s<T>(lw[K::O_W2T + m * 64 + lane], d2b, v4<T>{0, 0, 0, 0});
    f16_dact_tiles<T, MT>(a.act1, D1, H1);
    // ---- dW2[o][f] += sum_n delta2[n][o] H1[n][f]           (contracts over rows: U tiles)
    {
      const v4<T> d2u = f16_ld4(lw + K::O_D2 + (c & 3) * 16 + 4 * g);  // delta2[rows fi(g, .)][o = c] for c < 4 (zero for o >= dK)
#pragma unroll
      for (int n = 0; n < MT; ++n) {
        const v4<T> h1u = f16_ld4(lw + K::O_TB0 + (16 * n + c) * F16_TS + 4 * g);
        if constexpr (sizeof(T) == 8) {  // (Q44: not WIDE here)
          double q = dW2[n][0];
#pragma unroll
          for (int r = 0; r < 4; ++r) q = mfma44(d2u[r], h1u[r], q);
          dW2[n][0] = q;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) dW2[n] = mfma16s<T>(c < 4 ? d2u[r] : T(0), h1u[r], dW2[n]);
        }
      }
    }
    }
    f16_fence();  // H1^T has been read: its buffer takes delta1^T
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (!(F16_ABLATE & 8)) lw[K::O_TB0 + (16 * m + L::fi(g, r)) * F16_TS + pc] = D1[m][r];
    f16_fence();
    // ---- dH0 = delta1 W1 untransposed (A = delta1 T tiles with M = rows, B = theta's own W1 registers): U tiles
    v4<T> h0u[MT], d0u[MT];
    if constexpr ((V & 1) != 0) {
      // one hidden layer: "delta1" above already is delta0 (act1 = act0, H1 = H0); its U tiles come back from the buffer
#pragma unroll
      for (int n = 0; n < MT; ++n) d0u[n] = f16_ld4(lw + K::O_TB0 + (16 * n + c) * F16_TS + 4 * g);
    } else {
#pragma unroll
      for (int n = 0; n < MT; ++n) {
        h0u[n] = f16_ld4(lw + K::O_TB1 + (16 * n + c) * F16_TS + 4 * g);
        d0u[n] = v4<T>{0, 0, 0, 0};
      }
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int n = 0; n < MT; ++n) d0u[n] = mfma16<T>(D1[m][r], th[(m * MT + n) * 4 + r], d0u[n]);
      f16_dact_tiles<T, MT>(a.act0, d0u, h0u);
      // ---- dW1[out][in] += sum_n delta1[n][out] H0[n][in];  db1 += sum_n delta1
#pragma unroll
      for (int mo = 0; mo < MT; ++mo) {
        const v4<T> d1u = f16_ld4(lw + K::O_TB0 + (16 * mo + c) * F16_TS + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int n = 0; n < MT; ++n) dW1[mo * MT + n] = mfma16<T>(d1u[r], h0u[n][r], dW1[mo * MT + n]);
        db1[mo] += (d1u[0] + d1u[1]) + (d1u[2] + d1u[3]);
      }
    }
    // ---- dW0[out][in] += sum_n delta0[n][out] x[n][in];  db0 += sum_n delta0
    {
      T xu[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) xu[r] = (F16_ABLATE & 4) ? T(0.02 * lane) : xt[off_xu + r * 64 + lane];
      // (the 4x4x4 form was measured here too, for at most four inputs: x fetched with the input index modulo four and the
      // one result register moved into the slot layout once per evaluation by eight exchanges: +0.5 %, not kept)
#pragma unroll
      for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dW0[m] = mfma16s<T>(d0u[m][r], xu[r], dW0[m]);
        db0[m] += (d0u[m][0] + d0u[m][1]) + (d0u[m][2] + d0u[m][3]);
      }
    }
    f16_fence();
  }
  // ---- gather the gradient into the canonical registers
  if (GRAD) {
#pragma unroll
    for (int i = 0; i < MT * MT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) gr[i * 4 + r] = dW1[i][r];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        gr[K::S_W0 + m * 4 + r] = dW0[m][r];
        gr[K::S_W2 + m * 4 + r] = dW2[m][r];
      }
      gr[K::S_B1 + m] = f16_gsum(db1[m]);
      gr[K::S_B0 + m] = f16_gsum(db0[m]);
    }
    if ((V & 2) != 0 && a.dK > 4) {
      // the tile-layout sums over the rows: the 16 lanes of a lane group (DPP row reductions), total in every lane
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        T v = db2[r];
        v += f16_dpp<0xB1, 0xF>(v);
        v += f16_dpp<0x4E, 0xF>(v);
        v += f16_dpp<0x141, 0xF>(v);
        v += f16_dpp<0x140, 0xF>(v);
        gr[K::S_B2 + r] = v;
      }
    } else {
#pragma unroll
      for (int o = 0; o < 4; ++o) gr[K::S_B2 + o] = f16_wsum(db2[o]);
    }
  }
  // ---- prior (bayesian_model.py:46-50): elementwise Normal(mu, sigma); temperature scales everything (:33-34,48-49)
  T qsum = T(0);
  if (UPRIOR || a.prior_uniform) {
    const T mu0 = a.mu0, iv0 = a.iv0;
    F16_EACH(k) {
      const F16Slot s = f16_slot<H, T, V>(k, a, c, g, lane);
      const T d = th[k] - mu0;
      qsum += s.counts ? d * d * iv0 : T(0);
      if (GRAD) {
        T gn = gr[k] - d * iv0;
        if (has_temp) gn *= temp;
        gr[k] = s.valid ? gn : T(0);
      }
    }
  } else {
    F16_EACH(k) {  // (a slot that holds nothing reads element 0 and drops it: no load behind a per-lane branch)
      const F16Slot s = f16_slot<H, T, V>(k, a, c, g, lane);
      const int si = s.valid ? s.idx : 0;
      const T d = th[k] - a.mu[si];
      const T iv = a.inv_var[si];
      qsum += s.counts ? d * d * iv : T(0);
      if (GRAD) {
        T gn = gr[k] - d * iv;
        if (has_temp) gn *= temp;
        gr[k] = s.valid ? gn : T(0);
      }
    }
  }
  T prior = T(0);
  if (need_value) {
    lik = f16_wsum(lik);
    prior = a.prior_const - T(0.5) * f16_wsum(qsum);
  }
  if (has_temp) { lik *= temp; prior *= temp; }
  if (lik_out) *lik_out = lik;
  if (prior_out) *prior_out = prior;
  return lik + prior;
}

// Where a lane that holds no counted copy of an element sends its store of the final state (the rule of O_JUNK, for global
// memory: no store behind a per-lane branch while theta / momentum / gradient are live; every wave's lane l writes the same
// 8 bytes, nobody reads them).
static __device__ double g_f16_junk[64];
template <typename T>
__device__ __forceinline__ T* f16_junk(int lane) { return reinterpret_cast<T*>(g_f16_junk + lane); }

// One chain of one launch: everything between reading theta and writing the accepted state back.
template <typename T, int H, int V, int MODE, typename A>
__device__ __forceinline__ void f16_run_chain(const A& a, T* lw, const int64_t chain, const int it, const int c,
                                              const int g, const int lane) {
  typedef F16Cfg<H, (V & 2) ? 4 : 2> K;
  // later iterations of one launch read what this wave's lanes wrote at the end of the previous one
  if (it > 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const uint64_t iter = a.iter + (uint64_t)it;
  const int P = a.P;
  T* thg = a.theta + chain * P;
  T* grg = a.grad + chain * P;
  constexpr bool PLAIN = MODE == F16_HMC_PLAIN;
  const bool has_temp = !PLAIN && a.temp != nullptr;
  const T temp = has_temp ? a.temp[chain] : T(1);
  const T eps = a.step_vec ? a.step_vec[chain] : a.step;
#ifdef F16_ONLY_MODE  // diagnostic builds: one mode compiled in (a kernel a fifth of the size)
  const int mode = F16_ONLY_MODE;
#else
  const int mode = PLAIN ? (int)F16_HMC : (MODE >= 0 ? MODE : a.mode);  // MODE: the mode compiled in (EY_F16_HMC_OWN below), -1 = the argument's
#endif

  T th[K::NREG], gr[K::NREG];
  F16_EACH(k) {
    const F16Slot s = f16_slot<H, T, V>(k, a, c, g, lane);
    const T v = thg[s.valid ? s.idx : 0];
    th[k] = s.valid ? v : T(0);
    gr[k] = T(0);
  }

  if (mode == F16_GRAD) {
    f16_write_images<T, H, (V & 2) ? 4 : 2>(lw, th, a, c, g);
    T lik, prior;
    const T t = f16_eval<T, H, V>(a, lw, th, gr, a.grad != nullptr, has_temp, temp, c, g, lane, &lik, &prior);
    if (a.grad) {
      F16_EACH(k) {
        const F16Slot s = f16_slot<H, T, V>(k, a, c, g, lane);
        *(s.counts ? grg + s.idx : f16_junk<T>(lane)) = gr[k];
      }
    }
    if (lane == 0) {
      if (a.target) a.target[chain] = t;
      if (a.hcur) a.hcur[chain] = lik;     // ey_log_target: the two parts
      if (a.hprop) a.hprop[chain] = prior;
    }
    return;
  }

  // The chain's N(0,1) stream (or the caller's) for all P elements, indexed like theta, in the image region of this
  // wave's LDS (free between evaluations: P <= H^2 + 40 H): lane l computes the blocks of four l, l + 64, ... (one
  // Philox call per block), every lane then picks the elements of its register layout.
  T* st = lw + K::O_W1A;
  if (mode != F16_LEAPFROG) {
    if (a.p0) {
      const T* src = a.p0 + chain * P;
      for (int i = lane; i < P; i += 64) st[i] = src[i];
    } else {
      const EyRng rn = ey_rng_make(a.seed, a.chain_offset + (uint64_t)chain, iter, EY_STREAM_NORMAL);
      for (int b = lane; 4 * b < P; b += 64) {
        T o[4];
        ey_rng_normal4<T>(rn, (uint32_t)b, o);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (4 * b + j < P) st[4 * b + j] = o[j];
      }
    }
    f16_fence();
  }

  T p[K::NREG];  // HMC: the momentum; MALA / MH: the proposal

  if (mode == F16_MALA || mode == F16_MH) {
    // MALA.draw (mala.py:46-82) / MetropolisHastings.draw (metropolis_hastings.py:41-73): one evaluation at the proposal
    const T sc = a.step_vec ? Nm<T>::sqrt(eps) : a.sqrt_step;  // scale = sqrt(step), mala.py:39
    T gp[K::NREG];
    T qf = T(0);
    F16_EACH(k) {
      const F16Slot s = f16_slot<H, T, V>(k, a, c, g, lane);
      const int si = s.valid ? s.idx : 0;
      gp[k] = T(0);
      const T zi = st[si];
      if (mode == F16_MALA) {
        const T gk = grg[si];
        gr[k] = s.valid ? gk : gr[k];
        const T loc = th[k] + T(0.5) * eps * gk;  // kernel_mean, mala.py:35-36
        const T pk = loc + sc * zi;
        const T d = pk - loc;
        p[k] = s.valid ? pk : T(0);
        qf += s.counts ? d * d : T(0);
      } else {
        const T pk = th[k] + a.scale[si] * zi;  // NormalKernel(theta, scale).sample()
        p[k] = s.valid ? pk : T(0);
      }
    }
    f16_fence();  // the staged normals have been read; the evaluation reuses that LDS
    f16_write_images<T, H, (V & 2) ? 4 : 2>(lw, p, a, c, g);
    const T tv = f16_eval<T, H, V>(a, lw, p, gp, mode == F16_MALA, has_temp, temp, c, g, lane);
    const T t_old = a.target[chain];
    T log_rate = tv - t_old;  // symmetric kernel: metropolis_hastings.py:50
    if (mode == F16_MALA) {
      T qb = T(0);
      F16_EACH(k) {
        const F16Slot s = f16_slot<H, T, V>(k, a, c, g, lane);
        const T d = th[k] - (p[k] + T(0.5) * eps * gp[k]);
        qb += s.counts ? d * d : T(0);
      }
      const T inv2v = T(1) / (T(2) * sc * sc);
      log_rate += (f16_wsum(qf) - f16_wsum(qb)) * inv2v;  // the -P log s - P/2 log 2pi terms cancel (mala.py:58-64)
    }
    const EyRng ru = ey_rng_make(a.seed, a.chain_offset + (uint64_t)chain, iter, EY_STREAM_UNIFORM);
    const T u = a.u ? a.u[chain] : ey_rng_uniform<T>(ru);
    const bool acc = Nm<T>::log(u) < log_rate;  // mala.py:66, metropolis_hastings.py:56
    T* so = a.rec_samples ? a.rec_samples + ((int64_t)it * a.C + chain) * P : nullptr;
    const bool acc_u = __builtin_amdgcn_readfirstlane((int)acc) != 0;  // (the same in every lane: a scalar branch)
    F16_EACH(k) {
      const F16Slot s = f16_slot<H, T, V>(k, a, c, g, lane);
      if (acc_u) {
        *(s.counts ? thg + s.idx : f16_junk<T>(lane)) = p[k];
        if (mode == F16_MALA) *(s.counts ? grg + s.idx : f16_junk<T>(lane)) = gp[k];
      }
      if (so) *(s.counts ? so + s.idx : f16_junk<T>(lane)) = acc ? p[k] : th[k];
    }
    if (lane == 0) {
      if (acc) a.target[chain] = tv;
      a.accepted[chain] = acc ? 1 : 0;
      if (a.rate) a.rate[chain] = log_rate;
      if (a.rec_targets) a.rec_targets[(int64_t)it * a.C + chain] = acc ? tv : t_old;
      if (a.rec_accepted) a.rec_accepted[(int64_t)it * a.C + chain] = acc ? 1 : 0;
      if (a.accept_count && acc) a.accept_count[chain] += 1;
    }
    if (a.n_iters > 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    return;
  }

  // ---- HMC.draw (hmc.py:126-156) / HMC.leapfrog (:100-124)
  T t_cur = T(0), kin = T(0);
  if (mode == F16_HMC) {
    F16_EACH(k) {
      const F16Slot s = f16_slot<H, T, V>(k, a, c, g, lane);
      const int si = s.valid ? s.idx : 0;
      const T zv = st[si], gv = grg[si];
      p[k] = s.valid ? zv : T(0);   // hmc.py:134
      kin += s.counts ? p[k] * p[k] : T(0);
      gr[k] = (s.valid && !a.recompute) ? gv : gr[k];
    }
    f16_fence();  // the staged normals have been read; the evaluations reuse that LDS
    kin = f16_wsum(kin);
    t_cur = a.target[chain];
  } else {
    const T* pin = a.pio + chain * P;
    F16_EACH(k) {
      const F16Slot s = f16_slot<H, T, V>(k, a, c, g, lane);
      const T pv = pin[s.valid ? s.idx : 0];
      p[k] = s.valid ? pv : T(0);
    }
  }
  const T h_cur = -t_cur + T(0.5) * kin;  // hmc.py:91-98,137
  T t = t_cur;
  // leapfrog (grad_potential = -grad); every lane updates the elements it holds, replicas included (same values).
  // Step 0 is the evaluation at the starting position (hmc.py:104): skipped when the cached gradient is used.
  const int k_first = (mode == F16_LEAPFROG || a.recompute) ? 0 : 1;
  if (k_first == 1) {
    F16_EACH(k) p[k] = p[k] + T(0.5) * eps * gr[k];
  }
#pragma unroll 1
  for (int kk = k_first; kk <= a.L; ++kk) {
    if (kk > 0) {
      F16_EACH(k) th[k] = th[k] + eps * p[k];
    }
    f16_write_images<T, H, (V & 2) ? 4 : 2>(lw, th, a, c, g);
    // (inside a trajectory only the gradient is consumed, hmc.py:108-121: the value-only work -- the logs of the
    // likelihood terms -- runs at the end point alone)
    t = f16_eval<T, H, V, PLAIN>(a, lw, th, gr, true, has_temp, temp, c, g, lane, nullptr, nullptr, kk == a.L);
    const T w = (kk > 0 && kk < a.L) ? eps : T(0.5) * eps;
    F16_EACH(k) p[k] = p[k] + w * gr[k];
  }

  if (mode == F16_LEAPFROG) {
    T* pout = a.pio + chain * P;
    F16_EACH(k) {
      const F16Slot s = f16_slot<H, T, V>(k, a, c, g, lane);
      *(s.counts ? thg + s.idx : f16_junk<T>(lane)) = th[k];
      *(s.counts ? pout + s.idx : f16_junk<T>(lane)) = -p[k];  // hmc.py:122
      *(s.counts ? grg + s.idx : f16_junk<T>(lane)) = gr[k];
    }
    if (lane == 0) a.target[chain] = t;
    return;
  }

  kin = T(0);
  F16_EACH(k) {
    const F16Slot s = f16_slot<H, T, V>(k, a, c, g, lane);
    kin += s.counts ? p[k] * p[k] : T(0);
  }
  kin = f16_wsum(kin);
  const T h_prop = -t + T(0.5) * kin;
  T rate = Nm<T>::exp(h_cur - h_prop);  // hmc.py:143-146
  if (rate > T(1)) rate = T(1);
  const EyRng ru = ey_rng_make(a.seed, a.chain_offset + (uint64_t)chain, iter, EY_STREAM_UNIFORM);
  const T u = a.u ? a.u[chain] : ey_rng_uniform<T>(ru);
  const bool acc = u < rate;  // strict <, NaN => reject (hmc.py:148)
  T* so = a.rec_samples ? a.rec_samples + ((int64_t)it * a.C + chain) * P : nullptr;
  const bool acc_u = __builtin_amdgcn_readfirstlane((int)acc) != 0;  // (the same in every lane: a scalar branch)
  F16_EACH(k) {
    const F16Slot s = f16_slot<H, T, V>(k, a, c, g, lane);
    if (so) {  // the state this chain is left in (chain_list.py:64-67)
      const T cur = thg[s.counts ? s.idx : 0];
      *(s.counts ? so + s.idx : f16_junk<T>(lane)) = acc ? th[k] : cur;
    }
    if (acc_u) {
      *(s.counts ? thg + s.idx : f16_junk<T>(lane)) = th[k];
      *(s.counts ? grg + s.idx : f16_junk<T>(lane)) = gr[k];
    }
  }
  if (lane == 0) {
    if (acc) a.target[chain] = t;
    a.accepted[chain] = acc ? 1 : 0;
    if (a.rate) a.rate[chain] = rate;
    if (a.hcur) a.hcur[chain] = h_cur;
    if (a.hprop) a.hprop[chain] = h_prop;
    if (!PLAIN && a.da_state && it < a.da_n)  // the tuner step of hmc.py:158-163, per chain, without leaving the launch
      a.da_step[chain] = (T)ey_da_update(a.da_state + 3 * chain, a.da_tab + 3 * it, (double)rate, a.da_d,
                                         a.da_has_eub != 0, a.da_logeub, it == a.da_final_it);
    if (a.rec_targets) a.rec_targets[(int64_t)it * a.C + chain] = acc ? t : t_cur;
    if (a.rec_accepted) a.rec_accepted[(int64_t)it * a.C + chain] = acc ? 1 : 0;
    if (a.accept_count && acc) a.accept_count[chain] += 1;
  }
  if (a.n_iters > 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
}

// Persistent launch: one WAVES-wave workgroup per CU, every wave walks over chains blockIdx + gridDim * wave, ... and
// takes each through all iterations of the launch (chain-major, as ey_mfma32.hip).
template <typename T, int H, int WAVES, int V, int MODE = -1>
__global__ void __launch_bounds__(WAVES * 64, (WAVES + 3) / 4) k_fused16(F16Args<T> a) {
  typedef F16Cfg<H, (V & 2) ? 4 : 2> K;
  extern __shared__ __attribute__((aligned(32))) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15, g = lane >> 4;
  T* lw = smem + (size_t)wave * K::WAVE_ELEMS;
  // the zero padding of the operand images (skinny dimensions padded to a tile) and of the delta2 buffer
  for (int i = lane; i < K::O_B0 - K::O_W0A; i += 64) lw[K::O_W0A + i] = T(0);
  lw[K::O_D2 + lane] = T(0);
  f16_fence();
  const int64_t first = (int64_t)blockIdx.x + (int64_t)gridDim.x * wave, stride = (int64_t)gridDim.x * WAVES;
  const int n_iters = (a.mode == F16_HMC || a.mode == F16_MALA || a.mode == F16_MH) ? a.n_iters : 1;
  typedef const __attribute__((address_space(4))) F16Args<T> KA;
  for (int64_t chain = first; chain < a.C; chain += stride)
    for (int it = 0; it < n_iters; ++it) {
      // the arguments are read in place from the kernarg segment in every round: hoisted out of these loops they would
      // all stay live in scalar registers for the whole kernel and spill into vector registers
      KA* ap = (KA*)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(ap));
      f16_run_chain<T, H, V, MODE>(*ap, lw, chain, it, c, g, lane);
    }
}

#if EY_F16_PART == 0
// ----------------------------------------------------------------------------------------------- host side
// Operand-order data image, per 16-row tile: [ks0][64] x as the B operand of F0 (lane (row, in & 3), k-step in >> 2),
// [4][64] x as the B operand of dW0 (lane (in, row >> 2), step row & 3), [64] labels (-1 = padding row), [4][16] y.
template <typename T>
__global__ void k_f16_pack(const T* __restrict__ x, const T* __restrict__ y, const int* __restrict__ labels, int N,
                           int d0, int dK, int lik, int ntiles, int ks0, int xt_stride, T* __restrict__ img) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ntiles * 64) return;
  const int t = i >> 6, lane = i & 63, c = lane & 15, g = lane >> 4;
  T* xt = img + (size_t)t * xt_stride;
  const int off_xu = ks0 * 64, off_lab = off_xu + 256, off_y = off_lab + 64;
  for (int s = 0; s < ks0; ++s) {
    const int n = 16 * t + c, in = 4 * s + g;
    xt[s * 64 + lane] = (n < N && in < d0) ? x[(size_t)n * d0 + in] : T(0);
  }
  for (int r = 0; r < 4; ++r) {  // the rows of a U tile: fi(g, r)
    const int n = 16 * t + Lay<T>::fi(g, r);
    xt[off_xu + r * 64 + lane] = (n < N && c < d0) ? x[(size_t)n * d0 + c] : T(0);
  }
  {
    const int n = 16 * t + c;
    T lab = T(-1);
    if (n < N) lab = lik == EY_LIK_CE_SUM ? (T)labels[n] : T(0);
    xt[off_lab + lane] = lab;
    xt[off_y + lane] = (n < N && g < dK) ? y[(size_t)n * dK + g] : T(0);
  }
}

bool ey_fused16_supports(const ey_plan* pl) {
  const EyModel& m = pl->m;
  if (m.nl != 3 && m.nl != 2) return false;
  const int K = m.nl;              // two hidden layers, or one (the middle layer of the kernel is then skipped)
  const int h1 = m.dims[1], h2 = m.dims[K - 1];
  const int H = std::max(h1, h2);  // the tile grid is the next of 16 / 32 / 64; narrower layers are padded
  if (h1 < 1 || h2 < 1 || H > 64) return false;
  // padded tiles are worth it while the padded W1 is at most eight times the real one (MLP(2-3-2-1) on a 16 x 16 grid
  // would do 43 times its work: measured half the generic kernel's rate at 65 536 chains); with one hidden layer the
  // same bound on the width squared
  const int Hp = H <= 16 ? 16 : (H <= 32 ? 32 : 64);
  if (8 * h1 * h2 < Hp * Hp) return false;
  // up to 16 outputs under CE-sum (the padded instantiations' whole delta2 tile), up to 4 under BCE-sum
  if (m.dims[0] < 1 || m.dims[0] > 16 || m.dims[K] < 1 || m.dims[K] > (m.lik == EY_LIK_CE_SUM ? 16 : 4)) return false;
  for (int l = 0; l < K - 1; ++l)
    if (m.act[l] != EY_ACT_SIGMOID && m.act[l] != EY_ACT_TANH && m.act[l] != EY_ACT_RELU) return false;
  if (m.lik == EY_LIK_CE_SUM && m.act[K - 1] != EY_ACT_NONE) return false;
  if (m.lik == EY_LIK_BCE_SUM && m.act[K - 1] != EY_ACT_SIGMOID) return false;
  if (pl->dtype == EY_F64 && H > 32) return false;  // theta and the gradient alone would take 428 of 512 registers
  return true;
}

static int f16_xt_stride(int ks0) { return ks0 * 64 + 256 + 64 + 64; }

int ey_fused16_set_data(ey_plan* pl, hipStream_t s) {
  const EyModel& m = pl->m;
  const int ntiles = (m.N + 15) / 16, ks0 = (m.dims[0] + 3) / 4, xt = f16_xt_stride(ks0);
  const size_t es = pl->dtype == EY_F32 ? 4 : 8;
  const size_t need = es * (size_t)ntiles * xt;
  if (need > pl->xpack16_bytes) {
    EY_HIP(hipDeviceSynchronize());
    (void)hipFree(pl->d_xpack16);
    pl->d_xpack16 = nullptr;
    pl->xpack16_bytes = 0;
    EY_HIP(hipMalloc(&pl->d_xpack16, need));
    pl->xpack16_bytes = need;
  }
  const dim3 grid((ntiles * 64 + 255) / 256);
  if (es == 4)
    hipLaunchKernelGGL(k_f16_pack<float>, grid, dim3(256), 0, s, (const float*)pl->d_x, (const float*)pl->d_y,
                       (const int*)pl->d_labels, m.N, m.dims[0], m.dims[m.nl], m.lik, ntiles, ks0, xt, (float*)pl->d_xpack16);
  else
    hipLaunchKernelGGL(k_f16_pack<double>, grid, dim3(256), 0, s, (const double*)pl->d_x, (const double*)pl->d_y,
                       (const int*)pl->d_labels, m.N, m.dims[0], m.dims[m.nl], m.lik, ntiles, ks0, xt, (double*)pl->d_xpack16);
  EY_HIP(hipGetLastError());
  return EY_OK;
}

#endif  // EY_F16_PART == 0

template <typename T, int H, int WAVES, int V, int MODE>
static int f16_launch_kernel(F16Args<T>& a, unsigned grid, size_t bytes, hipStream_t s) {
  EY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fused16<T, H, WAVES, V, MODE>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  hipLaunchKernelGGL((k_fused16<T, H, WAVES, V, MODE>), dim3(grid), dim3(WAVES * 64), bytes, s, a);
  EY_HIP(hipGetLastError());
  return EY_OK;
}
// (defined in ey_fused16_d32.hip, EY_F16_PART 1)
int ey_f16_launch_d32(F16Args<double>& a, int n_cu, hipStream_t s);
int ey_f16_launch_f32h64_hmc(F16Args<float>& a, int v, unsigned grid, size_t bytes, hipStream_t s);
// (defined in ey_fused16_plain.hip, EY_F16_PART 2: the plain HMC kernels of the shapes at two and four waves per SIMD)
int ey_f16_launch_plain(F16Args<float>& a, int h, int v, unsigned grid, size_t bytes, hipStream_t s);
int ey_f16_launch_plain(F16Args<double>& a, int h, int v, unsigned grid, size_t bytes, hipStream_t s);

template <typename T, int H, int WAVES, int V>
static int f16_launch_t(F16Args<T>& a, int n_cu, hipStream_t s) {
#ifdef F16_ONLY_H  // diagnostic builds (tools/f16_bisect.sh): one instantiation only, the others refuse
  if constexpr (!(sizeof(T) == F16_ONLY_SIZE && H == F16_ONLY_H && V == F16_ONLY_V)) {
    ey_set_error("fused16: instantiation compiled out of this diagnostic build");
    return EY_ERR_UNSUPPORTED;
  } else
#endif
  {
  const size_t bytes = sizeof(T) * (size_t)WAVES * F16Cfg<H, (V & 2) ? 4 : 2>::WAVE_ELEMS;
  const unsigned grid = (unsigned)std::min<int64_t>(a.C, n_cu > 0 ? n_cu : 256);
  // the HMC draw of the one-wave-per-SIMD kernels as an instantiation of its own: with the other modes' code and live
  // ranges out of the way the kernel spills a third less (scratch 1804 -> 1252 bytes per lane on the f64 headline model's
  // instantiation, +3.7 %, same bits).  Those of the f32 H = 64 shapes live in ey_fused16_d32.hip as well (EY_F16_PART).
  if constexpr (WAVES <= EY_F16_HMC_OWN) if (a.mode == F16_HMC) {
#if EY_F16_PART == 0
    if constexpr (sizeof(T) == 4 && H == 64 && WAVES == 4) return ey_f16_launch_f32h64_hmc(a, V, grid, bytes, s);
    else
#endif
    {
      // ... and once more with the draw's run-time options compiled out (F16_HMC_PLAIN: the f64 headline instantiation then
      // spills 740 instead of 1224 bytes per lane, +2.2 %, same bits)
      if (a.prior_uniform && !a.temp && !a.da_state) return f16_launch_kernel<T, H, WAVES, V, F16_HMC_PLAIN>(a, grid, bytes, s);
      return f16_launch_kernel<T, H, WAVES, V, F16_HMC>(a, grid, bytes, s);
    }
  }
#if EY_F16_PART == 0 && EY_F16_PLAIN_UNIT
  // the shapes at two and four waves per SIMD: the plain HMC draw alone gets an instantiation (ey_fused16_plain.hip), +4-5 %
  if constexpr (WAVES > EY_F16_HMC_OWN)
    if (a.mode == F16_HMC && a.prior_uniform && !a.temp && !a.da_state) return ey_f16_launch_plain(a, H, V, grid, bytes, s);
#endif
  return f16_launch_kernel<T, H, WAVES, V, -1>(a, grid, bytes, s);
  }
}
// The variant V = TWO | 2 PAD is a template parameter: one hidden layer (the middle layer skipped) and hidden widths
// below the tile grid are separate instantiations -- as run-time flags they cost the exact two-hidden-layer shapes
// 1-12 % and up to 4.5 % (A/B of whole-library builds in one session, tools/ab_fused16.py).
template <typename T, int H, int WAVES>
static int f16_launch_w(F16Args<T>& a, int n_cu, hipStream_t s) {
  // (the exact 64-wide shape also takes the PAD instantiation: its register allocation happens to come out 19 % faster,
  // 8.7e6 against 7.3e6 leapfrog-steps/s x chains on MLP(4-64-64-3), same session)
  // (more than 8 inputs: the padded forms' four k-steps; a layer without a bias: its slots are padding slots)
  const bool pad = a.h1 != H || a.h2 != H || H == 64 || a.d0 > 8 || a.iB0 < 0 || a.iB2 < 0 || (!a.two && a.iB1 < 0) || a.dK > 4;
  if constexpr (H == 64) return a.two ? f16_launch_t<T, H, WAVES, 3>(a, n_cu, s) : f16_launch_t<T, H, WAVES, 2>(a, n_cu, s);
  else switch ((a.two ? 1 : 0) | (pad ? 2 : 0)) {
    case 0: return f16_launch_t<T, H, WAVES, 0>(a, n_cu, s);
    case 1: return f16_launch_t<T, H, WAVES, 1>(a, n_cu, s);
    case 2: return f16_launch_t<T, H, WAVES, 2>(a, n_cu, s);
    default: return f16_launch_t<T, H, WAVES, 3>(a, n_cu, s);
  }
}

#if EY_F16_PART == 2
template <typename T, int H, int WAVES>
static int f16_plain_v(F16Args<T>& a, int v, unsigned grid, size_t bytes, hipStream_t s) {
  switch (v) {
    case 0: return f16_launch_kernel<T, H, WAVES, 0, F16_HMC_PLAIN>(a, grid, bytes, s);
    case 1: return f16_launch_kernel<T, H, WAVES, 1, F16_HMC_PLAIN>(a, grid, bytes, s);
    case 2: return f16_launch_kernel<T, H, WAVES, 2, F16_HMC_PLAIN>(a, grid, bytes, s);
    default: return f16_launch_kernel<T, H, WAVES, 3, F16_HMC_PLAIN>(a, grid, bytes, s);
  }
}
int ey_f16_launch_plain(F16Args<float>& a, int h, int v, unsigned grid, size_t bytes, hipStream_t s) {
  return h == 16 ? f16_plain_v<float, 16, EY_F16_W16>(a, v, grid, bytes, s) : f16_plain_v<float, 32, 8>(a, v, grid, bytes, s);
}
int ey_f16_launch_plain(F16Args<double>& a, int h, int v, unsigned grid, size_t bytes, hipStream_t s) {
  return f16_plain_v<double, 16, 8>(a, v, grid, bytes, s);
}
#elif EY_F16_PART == 1
int ey_f16_launch_d32(F16Args<double>& a, int n_cu, hipStream_t s) { return f16_launch_w<double, 32, 4>(a, n_cu, s); }
int ey_f16_launch_f32h64_hmc(F16Args<float>& a, int v, unsigned grid, size_t bytes, hipStream_t s) {
  // (H = 64 only ever takes the padded instantiations, f16_launch_w)
  if (a.prior_uniform && !a.temp && !a.da_state)
    return v == 3 ? f16_launch_kernel<float, 64, 4, 3, F16_HMC_PLAIN>(a, grid, bytes, s)
                  : f16_launch_kernel<float, 64, 4, 2, F16_HMC_PLAIN>(a, grid, bytes, s);
  return v == 3 ? f16_launch_kernel<float, 64, 4, 3, F16_HMC>(a, grid, bytes, s)
                : f16_launch_kernel<float, 64, 4, 2, F16_HMC>(a, grid, bytes, s);
}
#else

// EY_F16_W16: waves per CU of the f32 H = 16 instantiations: sixteen (four per SIMD, 128 registers) are 13 - 16 % faster
// than eight on the small shapes (MLP(4-16-16-3) 34.4 -> 39.0 TFLOP/s, same bits).  At that register budget the kernel
// spills, and that is what exposed the family's per-lane branches (see O_JUNK): until they were removed the padded
// instantiation's MALA log-rate came out wrong there (profiles/r04_f16_w16_flags.txt).
template <typename T>
static int f16_launch(ey_plan* pl, F16Args<T>& a, hipStream_t s) {
  const EyModel& m = pl->m;
  const int K = m.nl;
  a.two = K == 2;
  a.d0 = m.dims[0]; a.dK = m.dims[K]; a.act0 = m.act[0]; a.act1 = a.two ? m.act[0] : m.act[1]; a.lik = m.lik; a.P = m.P;
  a.iW0 = m.woff[0]; a.iB0 = m.boff[0]; a.iW1 = a.two ? 0 : m.woff[1]; a.iB1 = a.two ? 0 : m.boff[1];  // (-1: no bias)
  a.iW2 = m.woff[K - 1]; a.iB2 = m.boff[K - 1];
  a.xpack = (const T*)pl->d_xpack16;
  a.ntiles = (m.N + 15) / 16;
  a.ks0 = (m.dims[0] + 3) / 4;
  a.xt = f16_xt_stride(a.ks0);
  a.mu = (const T*)m.mu;
  a.inv_var = (const T*)m.inv_var;
  a.prior_uniform = pl->prior_uniform ? 1 : 0;
  a.mu0 = (T)pl->prior_mu0;
  a.iv0 = (T)pl->prior_iv0;
  a.prior_const = (T)m.prior_const;
  a.h1 = m.dims[1]; a.h2 = m.dims[K - 1];
  const int H = std::max(a.h1, a.h2);
  // waves per CU by what the per-wave LDS region and the register file allow
  if constexpr (sizeof(T) == 4) {
    if (H <= 16) return f16_launch_w<float, 16, EY_F16_W16>(a, pl->n_cu, s);
    if (H <= 32) return f16_launch_w<float, 32, 8>(a, pl->n_cu, s);
    return f16_launch_w<float, 64, 4>(a, pl->n_cu, s);
  } else {
    if (H <= 16) return f16_launch_w<double, 16, 8>(a, pl->n_cu, s);
    return ey_f16_launch_d32(a, pl->n_cu, s);
  }
}

template <typename T>
static void f16_set_run(F16Args<T>& a, const EyRun* run) {
  a.n_iters = 1;
  if (run) {
    a.n_iters = run->n_iters;
    a.rec_samples = (T*)run->samples;
    a.rec_targets = (T*)run->targets;
    a.rec_accepted = (unsigned char*)run->accepted;
    a.accept_count = run->accept_count;
  }
}

template <typename T>
static int f16_hmc_t(ey_plan* pl, void* theta, void* target, void* grad, const void* p0, const void* u, double step,
                     const void* step_vec, int L, const void* temp, int64_t C, uint64_t seed, uint64_t iter,
                     uint64_t chain_offset, uint32_t flags, void* accepted, void* rate, void* hcur, void* hprop,
                     hipStream_t s, const EyRun* run, const EyDA* da) {
  F16Args<T> a = {};
  a.mode = F16_HMC;
  if (da && da->state) {
    a.da_state = da->state; a.da_tab = da->table; a.da_step = (T*)da->step; a.da_n = da->n;
    a.da_final_it = da->final_it; a.da_has_eub = da->has_eub; a.da_d = da->d; a.da_logeub = da->logeub;
  }
  // while a dual averaging is attached its step vector is THE step, also once its table is used up (include/eeyore_amd.h)
  if (da && da->step) step_vec = da->step;
  a.C = C; a.theta = (T*)theta; a.target = (T*)target; a.grad = (T*)grad; a.p0 = (const T*)p0; a.u = (const T*)u;
  a.step = (T)step; a.step_vec = (const T*)step_vec; a.L = L; a.temp = (const T*)temp; a.seed = seed; a.iter = iter;
  a.chain_offset = chain_offset; a.recompute = (flags & EY_RECOMPUTE_INITIAL_GRAD) ? 1 : 0;
  a.accepted = (unsigned char*)accepted; a.rate = (T*)rate; a.hcur = (T*)hcur; a.hprop = (T*)hprop;
  f16_set_run(a, run);
  return f16_launch<T>(pl, a, s);
}
int ey_fused16_hmc(ey_plan* pl, void* theta, void* target, void* grad, const void* p0, const void* u, double step,
                   const void* step_vec, int L, const void* temp, int64_t C, uint64_t seed, uint64_t iter,
                   uint64_t chain_offset, uint32_t flags, void* accepted, void* rate, void* hcur, void* hprop,
                   hipStream_t s, const EyRun* run, const EyDA* da) {
  if (pl->dtype == EY_F32)
    return f16_hmc_t<float>(pl, theta, target, grad, p0, u, step, step_vec, L, temp, C, seed, iter, chain_offset, flags,
                            accepted, rate, hcur, hprop, s, run, da);
  return f16_hmc_t<double>(pl, theta, target, grad, p0, u, step, step_vec, L, temp, C, seed, iter, chain_offset, flags,
                           accepted, rate, hcur, hprop, s, run, da);
}

template <typename T>
static int f16_mala_mh_t(ey_plan* pl, int mode, void* theta, void* target, void* grad, const void* z, const void* u,
                         double step, const void* step_vec, const void* scale, const void* temp, int64_t C,
                         uint64_t seed, uint64_t iter, uint64_t chain_offset, void* accepted, void* log_rate,
                         hipStream_t s, const EyRun* run) {
  F16Args<T> a = {};
  a.mode = mode;
  a.C = C; a.theta = (T*)theta; a.target = (T*)target; a.grad = (T*)(grad ? grad : theta); a.p0 = (const T*)z;
  a.u = (const T*)u; a.step = (T)step; a.sqrt_step = (T)sqrt(step); a.step_vec = (const T*)step_vec;
  a.scale = (const T*)scale; a.temp = (const T*)temp; a.seed = seed; a.iter = iter; a.chain_offset = chain_offset;
  a.accepted = (unsigned char*)accepted; a.rate = (T*)log_rate;
  f16_set_run(a, run);
  return f16_launch<T>(pl, a, s);
}
int ey_fused16_mala(ey_plan* pl, void* theta, void* target, void* grad, const void* z, const void* u, double step,
                    const void* step_vec, const void* temp, int64_t C, uint64_t seed, uint64_t iter,
                    uint64_t chain_offset, void* accepted, void* log_rate, hipStream_t s, const EyRun* run) {
  if (pl->dtype == EY_F32)
    return f16_mala_mh_t<float>(pl, F16_MALA, theta, target, grad, z, u, step, step_vec, nullptr, temp, C, seed, iter,
                                chain_offset, accepted, log_rate, s, run);
  return f16_mala_mh_t<double>(pl, F16_MALA, theta, target, grad, z, u, step, step_vec, nullptr, temp, C, seed, iter,
                               chain_offset, accepted, log_rate, s, run);
}
int ey_fused16_mh(ey_plan* pl, void* theta, void* target, const void* z, const void* u, const void* scale,
                  const void* temp, int64_t C, uint64_t seed, uint64_t iter, uint64_t chain_offset, void* accepted,
                  void* log_rate, hipStream_t s, const EyRun* run) {
  if (pl->dtype == EY_F32)
    return f16_mala_mh_t<float>(pl, F16_MH, theta, target, nullptr, z, u, 0.0, nullptr, scale, temp, C, seed, iter,
                                chain_offset, accepted, log_rate, s, run);
  return f16_mala_mh_t<double>(pl, F16_MH, theta, target, nullptr, z, u, 0.0, nullptr, scale, temp, C, seed, iter,
                               chain_offset, accepted, log_rate, s, run);
}

template <typename T>
static int f16_grad_t(ey_plan* pl, const void* theta, const void* temp, int64_t C, void* lik, void* prior, void* target,
                      void* grad, hipStream_t s) {
  F16Args<T> a = {};
  a.mode = F16_GRAD;
  a.C = C; a.theta = (T*)theta; a.target = (T*)target; a.grad = (T*)grad; a.temp = (const T*)temp;
  a.hcur = (T*)lik; a.hprop = (T*)prior;
  return f16_launch<T>(pl, a, s);
}
// target / grad (either may be null) and, for ey_log_target, the two parts lik / prior (either may be null)
int ey_fused16_log_target(ey_plan* pl, const void* theta, const void* temp, int64_t C, void* lik, void* prior,
                          void* target, void* grad, hipStream_t s) {
  if (pl->dtype == EY_F32) return f16_grad_t<float>(pl, theta, temp, C, lik, prior, target, grad, s);
  return f16_grad_t<double>(pl, theta, temp, C, lik, prior, target, grad, s);
}

template <typename T>
static int f16_leapfrog_t(ey_plan* pl, void* theta, void* p, double step, const void* step_vec, int L, const void* temp,
                          int64_t C, void* target, void* grad, hipStream_t s) {
  F16Args<T> a = {};
  a.mode = F16_LEAPFROG;
  a.C = C; a.theta = (T*)theta; a.pio = (T*)p; a.target = (T*)target; a.grad = (T*)grad; a.step = (T)step;
  a.step_vec = (const T*)step_vec; a.L = L; a.temp = (const T*)temp;
  return f16_launch<T>(pl, a, s);
}
int ey_fused16_leapfrog(ey_plan* pl, void* theta, void* p, double step, const void* step_vec, int L, const void* temp,
                        int64_t C, void* target, void* grad, hipStream_t s) {
  if (pl->dtype == EY_F32) return f16_leapfrog_t<float>(pl, theta, p, step, step_vec, L, temp, C, target, grad, s);
  return f16_leapfrog_t<double>(pl, theta, p, step, step_vec, L, temp, C, target, grad, s);
}
#endif  // EY_F16_PART == 0
