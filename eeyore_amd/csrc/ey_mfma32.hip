// Fused HMC trajectory kernel on the f32 matrix cores for MLP(4-32-32-3)-shaped models (BASELINE configs 3/4).
//
// One wavefront per chain; a persistent 8-wave workgroup per CU whose waves walk over chains (and, for ey_hmc_run,
// over iterations).  The whole draw -- momentum, L leapfrog steps with the MLP forward + hand-coded backward over all
// data rows, Hamiltonians, accept -- runs inside the launch; theta, momentum and the gradient never leave registers,
// weights are re-staged through LDS once per leapfrog step, activations once per 32-row tile.  HBM is touched at
// trajectory start and end only.
//
// Reference semantics restated (paths relative to papamarkou/eeyore): MLP.forward eeyore/models/mlp.py:45-50,
// CE-sum eeyore/constants/constants.py:17, log_target eeyore/models/bayesian_model.py:30-56, gradient
// eeyore/models/log_target_model.py:15-23 (autograd there), HMC.leapfrog / draw eeyore/samplers/hmc.py:100-156.
//
// Data layout ("T" layout): a 32x32 tile X[feature][row] lives in the 16 accumulator registers of
// v_mfma_f32_32x32x2_f32: lane (c = lane&31, h = lane>>5), register r = 4q+j  <->  X[feature = 8q+4h+j][row = c].
// Computing every layer transposed (H_l^T = W_l H_{l-1}^T) makes that accumulator tile directly the B operand of the
// next product when its k-step (q,j) assigns k-slot h to feature 8q+4h+j, so the forward chain F0 -> F1 and the
// backward chain dH1 -> dH0 need no data movement at all.  Only the weight-gradient products, which contract over
// the row index (the lane index of a T tile), need a transpose: the tile is written to LDS [feature][36] and read
// back with ds_read_b128 as lane <-> feature, 4 consecutive rows per read (conflict-free at stride 36).
// The 32->3 output layer and the 4->32 input-weight gradient use v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4
// outer products per instruction) so that the skinny dimension does not waste a 32-wide tile.
//
// Per-lane canonical registers of theta / momentum / gradient (29 floats each):
//   w1[4q+j] = W1[out = 8q+4h+j][in = c]   (the D layout of dW1 and the A operand of dH0 = W1^T delta1)
//   w0[i]    = W0[out = 4(c>>2)+i][in = c&3]  (D layout of the 4x4x1 product; both halves hold the same values)
//   w2[o]    = W2[o][k = c]                 (ditto)      b1 = b1[c], b0 = b0[c], b2[o] uniform
#include <algorithm>
#include <atomic>
#include <type_traits>

#include "ey_common.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MF_MAX_TILES 24  // 27 KB data image + 8 x 16.3 KB per-wave regions fit the 160 KB of a CU
// per-wave LDS carve, in floats (all offsets multiples of 4 => 16-byte aligned b128 accesses)
#define TS36 36
// The carve depends on BF3 (a compile-time constant in scope wherever these are used): the bf16x3 form keeps the three
// bf16 pieces of W1's rows (the A operand of the F1 product) in 6 private 16-byte slots per lane and needs the f32
// image of W1 only while those are made (it then lies in the first transpose buffer).
// BF3 = 2 (batches of at most MF_TRD_TILES row tiles: the data image leaves room for it): H0 and delta1 cross to the
// row-contracting product dW1 as the bf16 pieces they are split into anyway (the B operand of F1, the B operand of dH0),
// through two [piece][row][32 features] images of 6 KB each that ds_read_b64_tr_b16 reads transposed, in place of two f32
// round trips and two more splits.  The images take the transpose buffers' places (H1's transposed copy, dead by then,
// lies in the first 4.5 KB of delta1's image; delta0's, made when dW1 has read both, in H0's); the low pieces of W1's rows
// stay in registers (8), which makes the room.
// BF3 = 3 (the pipelined tile loop at one wave per SIMD, eval_pipe below: four waves share the CU's LDS, 37 KB each): two
// row tiles are in flight, so nothing shares a buffer any more -- W1 pieces 4 KB | H0's piece image, one per tile parity,
// 2 x 6 KB | delta1's piece image 6 KB | H1 transposed 4.5 KB | delta0 transposed 4.5 KB | small 2.5 KB.
#define O_W1P 0        // BF3: [piece 0..2][k-step 0..1][lane] x 16 bytes (BF3 >= 2: pieces 0..1)
#define O_W1IMG (BF3 == 3 ? 1024 : (BF3 == 2 ? 2560 : (BF3 ? 1536 : 0)))
#define O_TB0 (BF3 == 3 ? 4096 : (BF3 == 2 ? 2560 : (BF3 ? 1536 : 1152)))
#define O_TB1 (BF3 == 3 ? 1024 : (BF3 == 2 ? 1024 : (BF3 ? 2688 : 2304)))
#define O_SMALL (BF3 == 3 ? 7936 : (BF3 == 2 ? 4096 : (BF3 ? 3840 : 3456)))
#define O_STAGE (BF3 >= 2 ? O_TB1 : O_TB0)  // >= 2304 contiguous floats that are free between evaluations
#define O_H1T 5632     // BF3 = 3 only: H1 with lane <-> feature (for dW2), [32][36]
#define O_D0T 6784     // BF3 = 3 only: delta0 likewise (for dW0)
#define PIPE_IMG 1536  // floats of one piece image
#define O_W0IMG (O_SMALL + 0)     // [32][5]
#define O_W2IMG (O_SMALL + 160)   // [4][36]
#define O_W2TIMG (O_SMALL + 304)  // [32][4]
#define O_B0IMG (O_SMALL + 432)
#define O_B1IMG (O_SMALL + 464)
#define O_D2BUF (O_SMALL + 496)   // [4][2][16], the four outputs D2S floats apart
// Stride between the four outputs (delta2) / inputs (x) of the regrouped images: the ds_read_b128 of a lane group takes
// its four 16-byte pieces from jj = 0..3, which 32 floats apart land on the same banks for jj and jj + 2 (a two-way
// conflict on eight reads per tile: the 10 % SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of round 1); 36 apart they do not.
#define D2S 36
#define WAVE_FLOATS (BF3 == 3 ? 8576 : (BF3 == 2 ? 4736 : (BF3 ? 4480 : 4096)))
#define WAVE_FLOATS_OF(bf3) ((bf3) == 3 ? 8576 : ((bf3) == 2 ? 4736 : ((bf3) ? 4480 : 4096)))
#define XTILE_FLOATS 304  // per row tile: [32][5] (x0..x3, label) + [4][D2S] (x regrouped for the 4x4x1 product)

// canonical offsets of MLP(4-32-32-3) in theta
#define I_W0 0
#define I_B0 128
#define I_W1 160
#define I_B1 1184
#define I_W2 1216
#define I_B2 (1216 + 32 * DKV)   // DKV = outputs of the model: a compile-time constant in scope wherever these are used
#define NPAR (1216 + 33 * DKV)

// What varies between the models this kernel serves: MLP(4-32-32-DK), both hidden layers with activation ACT, head LIK.
//   CE-sum on DK = 3 logits (the headline model; eeyore/constants/constants.py:17) or BCE-sum on one sigmoid output
//   (eeyore/stats/loss.py:1-11, naive logs: NaN once the output saturates, which rejects).
template <int DK_, int ACT_, int LIK_>
struct MfShape {
  static constexpr int DK = DK_, ACT = ACT_, LIK = LIK_;
};
typedef MfShape<3, EY_ACT_SIGMOID, EY_LIK_CE_SUM> MfHeadline;

enum { MODE_HMC = 0, MODE_GRAD = 1, MODE_LEAPFROG = 2, MODE_MALA = 3, MODE_MH = 4 };

struct MfArgs {
  const float* xpack;  // [ntiles][288]
  const float* mu;
  const float* inv_var;
  int prior_uniform;   // all parameters share (mu0, iv0): no per-element prior loads
  float mu0, iv0;
  float prior_const;
  int ntiles;
  int short_last;      // 1: the last row tile holds at most 24 rows (its fourth 8-row k-group is all padding)
  int64_t C;
  float* theta;        // [C,P] in/out
  float* target;       // [C]
  float* grad;         // [C,P]
  const float* p0;     // HMC: [C,P] or null (Philox); LEAPFROG: unused
  float* pio;          // LEAPFROG: [C,P] in/out
  const float* u;      // [C] or null
  float step;
  float sqrt_step;     // MALA: sqrt(step) computed on the host in double (mala.py:39)
  const float* scale;  // MH: proposal scale [P]
  const float* step_vec;
  int L;
  const float* temp;
  uint64_t seed, iter, chain_offset;
  int recompute;
  unsigned char* accepted;
  float *rate, *hcur, *hprop;
  int balance;         // 1: the two waves of a SIMD keep in step through s_setprio (see Pace)
  int stagger;         // row tiles by which the second wave of a SIMD pair is kept AHEAD of the first (see Pace)
  double *mom_s1, *mom_s2, *mom_acc;  // attached running moments (ey_plan_attach_moments) or null
  // ey_hmc_run: n_iters consecutive draws per launch and the per-iteration records (each nullable)
  int n_iters;
  float* rec_samples;          // [n_iters, C, P]
  float* rec_targets;          // [n_iters, C]
  unsigned char* rec_accepted; // [n_iters, C]
  int* accept_count;           // [C], +=
  // attached per-chain dual averaging (ey_plan_attach_da): state [C,3], table rows of this launch's iterations
  double* da_state;
  const double* da_tab;
  float* da_step;              // [C]: the step of the next iteration (the same array step_vec reads)
  int da_n, da_final_it, da_has_eub;
  double da_d, da_logeub;
};

// The arguments as the kernels read them: in place in the kernarg segment (constant address space, scalar loads).
typedef const __attribute__((address_space(4))) MfArgs KArgs;

template <int DKV>
struct Vec {
  float w1[16];
  float w0[4];
  float w2[DKV];
  float b1, b0;
  float b2[DKV];
};

// Wave reductions without LDS traffic.  wsum: DPP adds inside each 16-lane row (quad swaps, half-mirror, mirror), then
// row_bcast:15 / row_bcast:31 carry the row totals up so that lane 63 holds the wave total, which v_readlane returns
// as a uniform value.  hsum(v) = v + (v of lane^32) in every lane: v_permlane32_swap_b32 on two copies of v leaves
// (low half, low half) and (high half, high half).  (Inline asm: this compiler's builtin drops the second result.)
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_get(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xF, false));
}
__device__ __forceinline__ float wsum(float v) {
  v += dpp_get<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v += dpp_get<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v += dpp_get<0x141, 0xF>(v);  // row_half_mirror
  v += dpp_get<0x140, 0xF>(v);  // row_mirror: every lane of a row holds the row total
  v += dpp_get<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
  v += dpp_get<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float hsum(float v) {
  float a = v, b = v;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
}
// The staged images of W0, b0, W1, b1 carry the factor -log2(e), so the accumulator holds -g*log2(e) and
// sigmoid(g) = 1 / (1 + 2^acc): one v_exp_f32, one v_add_f32, one v_rcp_f32 per element.
#define NEG_LOG2E (-1.4426950408889634f)
// EY_ABLATE (diagnostic builds only, tools/ablate.sh): 1 = no transpose stores, 2 = no 4x4x1 products,
// 4 = no transcendental in the sigmoid, 8 = no dW1 product.  Results are wrong in such a build; only its timing is read.
#ifndef EY_ABLATE
#define EY_ABLATE 0
#endif
// EY_PHASE_TIMING (diagnostic builds only, tools/phase_timing.py): per-phase s_memtime sums of the tile loop.
#ifndef EY_PHASE_TIMING
#define EY_PHASE_TIMING 0
#endif
#if EY_PHASE_TIMING
#if defined(EY_MF_PART) && EY_MF_PART == 1  // the pipelined form's unit keeps counters (and readers, at the end) of its own
#define g_ey_phase g_ey_phase_p
#define g_ey_dbg g_ey_dbg_p
#define g_ey_wave_t g_ey_wave_t_p
#endif
__device__ unsigned long long g_ey_phase[32];
__device__ int g_ey_dbg[64];
__device__ unsigned long long g_ey_wave_t[3 * 8192];  // per chain: kernel entry, wave start, wave end (s_memrealtime)
// s_memtime is served by one global unit (~2 ns per call over the whole chip): only one wave in 256 reads it
#define PH(i) do { if (ph_on) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); ph_acc[i] += n_ - ph_t; ph_t = n_; } } while (0)
#define KO(i) do { if (kt_on) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); ko_acc[i] += n_ - ko_t; ko_t = n_; } } while (0)
#else
// BF3 = 2: a scheduling barrier at the phase boundaries selected by EY_SB keeps the instruction scheduler from overlapping
// whole phases (it fills the 256 registers two waves per SIMD allow, and the allocator then spills).  After F0 and after
// F1: +1.5 % against none in two interleaved rounds of whole-library builds, the backward boundaries nothing.
#ifndef EY_SB
#define EY_SB 0x03
#endif
#define PH(i) do { if (BF3 == 2 && ((EY_SB >> (i)) & 1)) __builtin_amdgcn_sched_barrier(0); } while (0)
#define KO(i) do { } while (0)
#endif
// Packed f32 math (two elements per instruction at the rate of one): written on two-element vectors so that the
// instruction selector sees v2f32 operations (it scalarises the same operations on a whole 16-element tile).
__device__ __forceinline__ f32x2 pk_add1(f32x2 a) { return a + 1.0f; }
__device__ __forceinline__ f32x2 pk_h_one_minus_h(f32x2 h) { return __builtin_elementwise_fma(-h, h, h); }
__device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 b) { return a * b; }
// The staged images of W0, b0, W1, b1 carry the activation's factor, so the accumulator holds s * g:
//   sigmoid(g) = 1 / (1 + 2^(-log2e g))            s = -log2(e):   v_exp_f32, packed "+1", v_rcp_f32
//   tanh(g)    = 2 / (1 + 2^(-2 log2e g)) - 1      s = -2 log2(e): the same and one fma
//   relu(g)    = max(g, 0)                          s = 1
template <int ACT>
struct ActScale { static constexpr float value = ACT == EY_ACT_SIGMOID ? NEG_LOG2E : (ACT == EY_ACT_TANH ? 2.0f * NEG_LOG2E : 1.0f); };
template <int ACT>
__device__ __forceinline__ f32x16 act_tile(const f32x16& a) {
  f32x16 h;
  if (ACT == EY_ACT_RELU) {
#pragma unroll
    for (int r = 0; r < 16; ++r) h[r] = fmaxf(a[r], 0.0f);
    return h;
  }
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    f32x2 e;
    e[0] = (EY_ABLATE & 4) ? a[r] * 0.01f - 0.5f : __builtin_amdgcn_exp2f(a[r]);
    e[1] = (EY_ABLATE & 4) ? a[r + 1] * 0.01f - 0.5f : __builtin_amdgcn_exp2f(a[r + 1]);
    e = pk_add1(e);
    h[r] = (EY_ABLATE & 4) ? e[0] : __builtin_amdgcn_rcpf(e[0]);
    h[r + 1] = (EY_ABLATE & 4) ? e[1] : __builtin_amdgcn_rcpf(e[1]);
    if (ACT == EY_ACT_TANH) {
      h[r] = __builtin_fmaf(2.0f, h[r], -1.0f);
      h[r + 1] = __builtin_fmaf(2.0f, h[r + 1], -1.0f);
    }
  }
  return h;
}
// d * act'(g) from the activation value h:  sigmoid h (1 - h),  tanh 1 - h^2,  relu [h > 0]
template <int ACT>
__device__ __forceinline__ f32x16 times_dact(const f32x16& d, const f32x16& h) {
  f32x16 o;
  if (ACT == EY_ACT_RELU) {
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = h[r] > 0.0f ? d[r] : 0.0f;
    return o;
  }
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    const f32x2 h2 = {h[r], h[r + 1]}, d2 = {d[r], d[r + 1]};
    const f32x2 dh = ACT == EY_ACT_TANH ? __builtin_elementwise_fma(-h2, h2, f32x2{1.0f, 1.0f}) : pk_h_one_minus_h(h2);
    const f32x2 v = pk_mul(d2, dh);
    o[r] = v[0];
    o[r + 1] = v[1];
  }
  return o;
}
__device__ __forceinline__ float sigmoid_from_scaled(float a) {
  if (EY_ABLATE & 4) return a * 0.01f + 0.5f;
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a));
}
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  if (EY_ABLATE & 2) { c[0] += a * 1e-9f; return c; }
  return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}
// ---- bf16x3 form of the three 32x32x32 products (BF3): every f32 operand element is split into three bf16 pieces,
// x = hi + mid + lo EXACTLY (round-to-nearest pieces of 8 significant bits each: |mid| <= 2^-9 |x|, |lo| <= 2^-18 |x|), and
// a b is summed from the six piece products of relative size >= 2^-18 on v_mfma_f32_32x32x16_bf16 (products of bf16 are
// exact in f32; accumulation in f32), smallest terms first.  The three dropped products are below 2^-26 |a b|.
// Measured against f64 (tools/bf16x3_probe.hip, profiles/r03_bf16x3_probe.txt): rms error 0.56 and maximum 6.2 in units of
// 2^-24 sum_k |a_k b_k| against 0.95 and 10.4 for the exact kernel's k-ordered f32 fma chain (v_mfma_f32_32x32x2_f32).
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct Pieces { u32x4 hi[2], mid[2], lo[2]; };  // [k-step s]: operand elements 8s .. 8s+7, packed in pairs
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));  // v_cvt_pk_bf16_f32, round to nearest even
}
__device__ __forceinline__ void split16(const f32x16& v, Pieces& P) {
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const float a = v[8 * s + 2 * d], b = v[8 * s + 2 * d + 1];
      const unsigned hh = pk_bf16(a, b);
      const float ra = a - __builtin_bit_cast(float, hh << 16), rb = b - __builtin_bit_cast(float, hh & 0xffff0000u);
      const unsigned mm_ = pk_bf16(ra, rb);
      const float sa = ra - __builtin_bit_cast(float, mm_ << 16), sb = rb - __builtin_bit_cast(float, mm_ & 0xffff0000u);
      P.hi[s][d] = hh;
      P.mid[s][d] = mm_;
      P.lo[s][d] = pk_bf16(sa, sb);
    }
}
// the same split of one pair in two steps (the pipelined tile loop places them in different MFMA gaps)
__device__ __forceinline__ unsigned split_hi(float a, float b, float& ra, float& rb) {
  const unsigned hh = pk_bf16(a, b);
  ra = a - __builtin_bit_cast(float, hh << 16);
  rb = b - __builtin_bit_cast(float, hh & 0xffff0000u);
  return hh;
}
__device__ __forceinline__ unsigned split_mid_lo(float ra, float rb, unsigned& ll) {
  const unsigned mm_ = pk_bf16(ra, rb);
  const float sa = ra - __builtin_bit_cast(float, mm_ << 16), sb = rb - __builtin_bit_cast(float, mm_ & 0xffff0000u);
  ll = pk_bf16(sa, sb);
  return mm_;
}
__device__ __forceinline__ f32x16 mfma_bf16(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// acc += A B over the 32 contracted indices both operands hold as elements 8s+j of k-step s
// (SWAPPED: the piece products in the order product_bf3(B, A) takes them, for a product whose operands changed sides:
// the same sums, term for term, as before the change)
template <bool SWAPPED = false>
__device__ __forceinline__ f32x16 product_bf3(const Pieces& A, const Pieces& B, f32x16 acc) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    if (SWAPPED) {
      acc = mfma_bf16(A.lo[s], B.hi[s], acc);
      acc = mfma_bf16(A.hi[s], B.lo[s], acc);
    } else {
      acc = mfma_bf16(A.hi[s], B.lo[s], acc);
      acc = mfma_bf16(A.lo[s], B.hi[s], acc);
    }
    acc = mfma_bf16(A.mid[s], B.mid[s], acc);
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    if (SWAPPED) {
      acc = mfma_bf16(A.mid[s], B.hi[s], acc);
      acc = mfma_bf16(A.hi[s], B.mid[s], acc);
    } else {
      acc = mfma_bf16(A.hi[s], B.mid[s], acc);
      acc = mfma_bf16(A.mid[s], B.hi[s], acc);
    }
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) acc = mfma_bf16(A.hi[s], B.hi[s], acc);
  return acc;
}

// the K-th of the twelve piece products of product_bf3<SWAPPED>, in its order
template <int K, bool SWAPPED>
__device__ __forceinline__ f32x16 bf3_step(const Pieces& A, const Pieces& B, const f32x16& acc) {
  if constexpr (K < 6) {
    constexpr int s = K / 3, j = K % 3;
    if constexpr (j == 0) return SWAPPED ? mfma_bf16(A.lo[s], B.hi[s], acc) : mfma_bf16(A.hi[s], B.lo[s], acc);
    else if constexpr (j == 1) return SWAPPED ? mfma_bf16(A.hi[s], B.lo[s], acc) : mfma_bf16(A.lo[s], B.hi[s], acc);
    else return mfma_bf16(A.mid[s], B.mid[s], acc);
  } else if constexpr (K < 10) {
    constexpr int s = (K - 6) / 2, j = (K - 6) % 2;
    if constexpr (j == 0) return SWAPPED ? mfma_bf16(A.mid[s], B.hi[s], acc) : mfma_bf16(A.hi[s], B.mid[s], acc);
    else return SWAPPED ? mfma_bf16(A.hi[s], B.mid[s], acc) : mfma_bf16(A.mid[s], B.hi[s], acc);
  } else {
    return mfma_bf16(A.hi[K - 10], B.hi[K - 10], acc);
  }
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>());
  }
}

__device__ __forceinline__ void wave_lds_fence() {
  // LDS operations of one wave execute in issue order; this only stops the compiler from moving them.
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// visit every canonical element a lane holds: f(value&, canonical index, counts) where `counts` says whether this
// lane's copy is the one that enters sums over parameters (replicas in the other half / other lanes do not)
template <int DKV, typename F>
__device__ __forceinline__ void for_each(Vec<DKV>& v, int c, int h, int lane, F f) {
#pragma unroll
  for (int r = 0; r < 16; ++r) f(v.w1[r], I_W1 + (8 * (r >> 2) + 4 * h + (r & 3)) * 32 + c, true);
#pragma unroll
  for (int i = 0; i < 4; ++i) f(v.w0[i], I_W0 + (4 * (c >> 2) + i) * 4 + (c & 3), h == 0);
#pragma unroll
  for (int o = 0; o < DKV; ++o) f(v.w2[o], I_W2 + o * 32 + c, h == 0);
  f(v.b1, I_B1 + c, h == 0);
  f(v.b0, I_B0 + c, h == 0);
#pragma unroll
  for (int o = 0; o < DKV; ++o) f(v.b2[o], I_B2 + o, lane == 0);
}
template <int DKV, typename F>
__device__ __forceinline__ void for_each_pair(Vec<DKV>& a, Vec<DKV>& b, int c, int h, int lane, F f) {
#pragma unroll
  for (int r = 0; r < 16; ++r) f(a.w1[r], b.w1[r], I_W1 + (8 * (r >> 2) + 4 * h + (r & 3)) * 32 + c, true);
#pragma unroll
  for (int i = 0; i < 4; ++i) f(a.w0[i], b.w0[i], I_W0 + (4 * (c >> 2) + i) * 4 + (c & 3), h == 0);
#pragma unroll
  for (int o = 0; o < DKV; ++o) f(a.w2[o], b.w2[o], I_W2 + o * 32 + c, h == 0);
  f(a.b1, b.b1, I_B1 + c, h == 0);
  f(a.b0, b.b0, I_B0 + c, h == 0);
#pragma unroll
  for (int o = 0; o < DKV; ++o) f(a.b2[o], b.b2[o], I_B2 + o, lane == 0);
}
template <int DKV, typename F>
__device__ __forceinline__ void for_each3(Vec<DKV>& a, Vec<DKV>& b, Vec<DKV>& d, int c, int h, int lane, F f) {
#pragma unroll
  for (int r = 0; r < 16; ++r) f(a.w1[r], b.w1[r], d.w1[r], I_W1 + (8 * (r >> 2) + 4 * h + (r & 3)) * 32 + c, true);
#pragma unroll
  for (int i = 0; i < 4; ++i) f(a.w0[i], b.w0[i], d.w0[i], I_W0 + (4 * (c >> 2) + i) * 4 + (c & 3), h == 0);
#pragma unroll
  for (int o = 0; o < DKV; ++o) f(a.w2[o], b.w2[o], d.w2[o], I_W2 + o * 32 + c, h == 0);
  f(a.b1, b.b1, d.b1, I_B1 + c, h == 0);
  f(a.b0, b.b0, d.b0, I_B0 + c, h == 0);
#pragma unroll
  for (int o = 0; o < DKV; ++o) f(a.b2[o], b.b2[o], d.b2[o], I_B2 + o, lane == 0);
}
template <int DKV, typename F>
__device__ __forceinline__ void for_each2(Vec<DKV>& a, Vec<DKV>& b, F f) {
#pragma unroll
  for (int r = 0; r < 16; ++r) f(a.w1[r], b.w1[r]);
#pragma unroll
  for (int i = 0; i < 4; ++i) f(a.w0[i], b.w0[i]);
#pragma unroll
  for (int o = 0; o < DKV; ++o) f(a.w2[o], b.w2[o]);
  f(a.b1, b.b1);
  f(a.b0, b.b0);
#pragma unroll
  for (int o = 0; o < DKV; ++o) f(a.b2[o], b.b2[o]);
}

// No memory operation of this kernel sits behind a per-lane branch (a register that the allocator spills and
// reloads inside one comes back with its inactive lanes lost).  Where only some lanes have something to store, the work is
// shared out instead: the two halves of the wave hold the same small vectors (W0, W2, b1, b0 and a row's delta2), so each
// half stores half of their images; replicas store the final state beside the counted copy (the same bits to the same
// address); what lane 0 alone published, every lane stores (or, in the tile loop, the other lanes store into a dead word: Pace::junk).
// Measured against the branches: same bits, 0.789 against 0.793 ms per draw (profiles/r04_mfma32_no_lane_branches.txt).
// stage the operand images of theta in this wave's LDS region
struct W1Lo { u32x4 v[2]; };  // BF3 = 2: the low pieces of W1's row c (the A operand of F1), kept in registers
template <int BF3, typename SH>
__device__ __forceinline__ void write_images(float* lw, const Vec<SH::DK>& th, int c, int h, int lane, W1Lo& wl) {
  constexpr float SC = ActScale<SH::ACT>::value;
#pragma unroll
  for (int r = 0; r < 16; ++r) lw[O_W1IMG + (8 * (r >> 2) + 4 * h + (r & 3)) * TS36 + c] = SC * th.w1[r];
  {
    // (no store behind a per-lane branch, and no store for nothing: the two halves of the wave hold the same small vectors,
    // so each half stores half of their images -- W0's rows i = 0, 1 | 2, 3; W2's image | its transpose; b1 | b0)
    const bool up = h != 0;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      lw[O_W0IMG + (4 * (c >> 2) + j + 2 * h) * 5 + (c & 3)] = SC * (up ? th.w0[j + 2] : th.w0[j]);
    const int w2base = up ? O_W2TIMG + c * 4 : O_W2IMG + c, w2step = up ? 1 : TS36;
#pragma unroll
    for (int o = 0; o < 4; ++o)  // outputs beyond DK are zero rows / columns of the two images
      lw[w2base + o * w2step] = o < SH::DK ? th.w2[o < SH::DK ? o : 0] : 0.0f;
    lw[(up ? O_B0IMG : O_B1IMG) + c] = SC * (up ? th.b0 : th.b1);
  }
  wave_lds_fence();
  if constexpr (BF3) {
    // row c of the scaled W1 in the element order of the F1 product's k-steps (element 4q+j <-> input 8q+4h+j, the order
    // in which the accumulator tile H0 holds its features), split once per position; the f32 image is dead after this
    f32x16 wr;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(lw + O_W1IMG + c * TS36 + 8 * q + 4 * h);
#pragma unroll
      for (int j = 0; j < 4; ++j) wr[4 * q + j] = wv[j];
    }
    Pieces A;
    split16(wr, A);
    u32x4* pv = reinterpret_cast<u32x4*>(lw + O_W1P) + lane;
    pv[0 * 64] = A.hi[0]; pv[1 * 64] = A.hi[1];
    pv[2 * 64] = A.mid[0]; pv[3 * 64] = A.mid[1];
    if constexpr (BF3 >= 2) {
      wl.v[0] = A.lo[0]; wl.v[1] = A.lo[1];
    } else {
      pv[4 * 64] = A.lo[0]; pv[5 * 64] = A.lo[1];
    }
    wave_lds_fence();
  }
}

__device__ __forceinline__ void store_T(float* tb, const f32x16& v, int c, int h) {
  if (EY_ABLATE & 1) { tb[c] = v[0] + v[5] + v[10] + v[15]; return; }
#pragma unroll
  for (int r = 0; r < 16; ++r) tb[(8 * (r >> 2) + 4 * h + (r & 3)) * TS36 + c] = v[r];
}

// Keeping the two waves of a SIMD in step.  The instruction arbiter serves the OLDER of two ready waves, so of two
// chains sharing a SIMD one runs at nearly its stand-alone speed and the other on what is left (measured with
// tools/wave_timeline.py: lifetimes 417 vs 632 us); at the end of the launch the late waves then run alone, and a wave
// alone uses the ALU far less than two do.  Each wave therefore publishes in LDS how many row tiles it still has to
// evaluate in this launch and, once per tile, compares with its SIMD partner: the wave with less left drops to
// priority 0, the other takes priority 1, so the two waves of a SIMD finish the launch together whatever their
// numbers of chains.  The partner's counter is read at the top of a tile and consumed in the middle of it, so the
// LDS latency is never waited for.
struct Pace {
  int* prog;    // [waves] tiles left (+ bias), per wave of this workgroup
  int* junk;    // a word nobody reads any more (this wave's SIMD id slot, dead once the partners are found)
  int wave, partner;
  int left;
  int bias;     // added to what this wave publishes and compares: the wave then runs `bias` tiles ahead of its partner
  bool on;
  bool hi;      // EY_PHASE_PRIO: this wave is the one behind (the low bit of its priority levels)
};
__device__ __forceinline__ int pace_post(Pace& pc, int lane) {
  if (!pc.on) return 0;
  pc.left = __builtin_amdgcn_readfirstlane(pc.left - 1);
  // (lane 0 publishes; the other lanes store into this wave's junk word: no store behind a per-lane branch)
  __hip_atomic_store(lane == 0 ? &pc.prog[pc.wave] : pc.junk, pc.left + pc.bias, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return __hip_atomic_load(&pc.prog[pc.partner], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// EY_PHASE_PRIO (diagnostic builds, default off): priority by phase, see chain_enter below
#ifndef EY_PHASE_PRIO
#define EY_PHASE_PRIO 0
#endif
__device__ __forceinline__ void pace_apply(Pace& pc, int theirs_v) {
  if (!pc.on) return;
  // (scalar comparison: a vector compare would be lowered to EXEC masking, under which both s_setprio would execute)
  const int theirs = __builtin_amdgcn_readfirstlane(theirs_v);
  if (EY_PHASE_PRIO == 1 || EY_PHASE_PRIO == 2) {  // the levels are set where the phases change (chain_enter / chain_exit)
    if (EY_PHASE_PRIO == 2) pc.hi = !(pc.left + pc.bias < theirs);
    return;
  }
  if (pc.left + pc.bias < theirs) __builtin_amdgcn_s_setprio(0);
  else __builtin_amdgcn_s_setprio(1);
}

// Priority by phase (EY_PHASE_PRIO = 1: levels 0 inside / 2 outside a chain of bf16 products, no pacing; 2: the pacing bit
// as the low bit of both levels; 3: only the scheduling barriers).  MEASURED AND NOT KEPT (round 4).  The probe
// (tools/coexec2_probe.hip, profiles/r04_coexec2_probe.txt) shows that a wave whose next instruction is a
// v_mfma_f32_32x32x16_bf16 waiting for the matrix pipe holds the SIMD's vector issue port -- beside a wave inside a chain of
// such products the partner's vector instructions do not issue at all, the two waves' times ADD -- unless the partner has
// the higher priority: then its plain / transcendental / convert instructions run in the MFMAs' shadow (16 MFMAs + 128
// v_fma_f32: 30.7 ns per MFMA slot at equal priority, 21.3 with the vector wave at priority 1; packed f32 instructions and
// f32 MFMAs never overlap with a bf16 MFMA, in either wave).  In this kernel it does not pay: levels by phase without the
// pacing 0.886 ms per draw against 0.800 (the older wave of each SIMD then finishes its chains in 396 us and the younger
// in 599, the tail of the launch runs one wave per SIMD; tools/wave_timeline.py), with the pacing bit below the phase bit
// 0.837, and 0.809 when the device code is also compiled without packed f32 instructions (profiles/r04_ab_phase_prio.txt).
// The paired throughput is the same with and without (4.19 against 4.17 chains per ms and SIMD): the vector phases are
// themselves 45 % f32 MFMA and packed instructions, which nothing overlaps.
#ifndef EY_PHASE_SB
#define EY_PHASE_SB 1
#endif
__device__ __forceinline__ void chain_enter(const Pace& pc) {
  if (EY_PHASE_PRIO == 0) return;
  if (EY_PHASE_SB) __builtin_amdgcn_sched_barrier(0);
  if (EY_PHASE_PRIO == 3) return;
  if (EY_PHASE_PRIO == 2 && pc.hi) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
}
__device__ __forceinline__ void chain_exit(const Pace& pc) {
  if (EY_PHASE_PRIO == 0) return;
  if (EY_PHASE_PRIO != 3) {
    if (EY_PHASE_PRIO == 2 && pc.hi) __builtin_amdgcn_s_setprio(3);
    else __builtin_amdgcn_s_setprio(2);
  }
  if (EY_PHASE_SB) __builtin_amdgcn_sched_barrier(0);
}

// ---- the piece images of BF3 = 2: a tile X[feature][row] in the T layout, already split, goes to LDS as
// [piece][row c][four 16-byte units] -- unit 2s+h holds features 16s+4h .. +3 and 16s+8+4h .. +3 of the row (the lane's
// elements 8s .. 8s+7), at position (2s+h) ^ ((c>>1)&3) so that eight consecutive rows of a store fill all banks -- and
// comes back with lane <-> feature: ds_read_b64_tr_b16 gives lane i of a 16-lane group feature 16(group&1) + i of the
// four rows 8s+4h+q whose addresses lanes 4q+p supply (features 16(group&1) + 4p .. +3): elements 4s .. 4s+3 of the
// lane <-> feature layout, packed in pairs as the products take them.  Conflict-free both ways.
__device__ __forceinline__ void store_pieces(float* img, const Pieces& P, int c, int h) {
  u32x4* pb = reinterpret_cast<u32x4*>(img) + c * 4;
  const int sw = (c >> 1) & 3;
  const int u0 = h ^ sw, u1 = (2 + h) ^ sw;
  pb[0 * 128 + u0] = P.hi[0]; pb[0 * 128 + u1] = P.hi[1];
  pb[1 * 128 + u0] = P.mid[0]; pb[1 * 128 + u1] = P.mid[1];
  pb[2 * 128 + u0] = P.lo[0]; pb[2 * 128 + u1] = P.lo[1];
}
__device__ __forceinline__ void load_pieces_transposed(const float* img, Pieces& P, int h, int lane) {
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const int q = (lane >> 2) & 3, pp = lane & 3, cb = (lane >> 4) & 1;
  const int n0 = 4 * h + q;
  const char* base = reinterpret_cast<const char*>(img) + n0 * 64 + 16 * ((2 * cb + (pp & 1)) ^ ((n0 >> 1) & 3)) + 8 * (pp >> 1);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const u32x2 wh = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + 0 * 2048 + 512 * s)));
    const u32x2 wm = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + 1 * 2048 + 512 * s)));
    const u32x2 wl = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + 2 * 2048 + 512 * s)));
    P.hi[s >> 1][2 * (s & 1)] = wh[0]; P.hi[s >> 1][2 * (s & 1) + 1] = wh[1];
    P.mid[s >> 1][2 * (s & 1)] = wm[0]; P.mid[s >> 1][2 * (s & 1) + 1] = wm[1];
    P.lo[s >> 1][2 * (s & 1)] = wl[0]; P.lo[s >> 1][2 * (s & 1) + 1] = wl[1];
  }
}

// EY_MF_GAPS: which bf16 products of the tile loop (BF3 = 2) take vector work of their own tile into their gaps
// (bit 0: dW1 <- delta0 and its transposed store).  See eval.  MEASURED AND NOT KEPT (round 5, profiles/r05_ab_gaps.txt):
// same bits, 0.8183 / 0.8154 against 0.8188 / 0.8157 ms per draw -- with two waves per SIMD the gaps of one wave's products
// are not what the pair waits for.
#ifndef EY_MF_GAPS
#define EY_MF_GAPS 0
#endif
// (an MFMA is a pure value to the instruction selector, which may emit it on the far side of any number of scheduling
// barriers; the empty volatile statement takes the accumulator in and out, which pins the product that made it)
#define GAP_PIN(x) asm volatile("" : "+v"(x))
// d * act'(g) from the activation value, one element, as times_dact computes it -- through opaque statements, so that the
// vectoriser does not pair the elements into packed instructions again
template <int ACT>
__device__ __forceinline__ float dact_gap(float d, float hh) {
  if (ACT == EY_ACT_RELU) return hh > 0.0f ? d : 0.0f;
  float dh, o;
  if (ACT == EY_ACT_TANH) asm("v_fma_f32 %0, -%1, %1, 1.0" : "=v"(dh) : "v"(hh));
  else asm("v_fma_f32 %0, -%1, %1, %1" : "=v"(dh) : "v"(hh));
  asm("v_mul_f32_e32 %0, %1, %2" : "=v"(o) : "v"(d), "v"(dh));
  return o;
}
// log-target and gradient of the position whose images are staged in lw.  Returns the (tempered) log-target.
// `need_value` (wave-uniform) = false skips the value-only work (row log-sum-exp terms, quadratic form of the prior
// and their reductions): inside a trajectory only the gradient is consumed, hmc.py:108-121.
//
// PARK: the twelve elements of the position that the tile loop does not read from registers (W0, W2, b0, b1 are
// staged as images; b2 is taken into scalar registers) wait in this wave's LDS region while the tile loop runs.  The
// loop needs every vector register it can get, and what does not fit is spilled to scratch memory, whose reload
// latency the end of every evaluation then waits for (four serial round trips per leapfrog step before this).
// GRAD = false: the value only (random-walk MH needs no gradient, metropolis_hastings.py:41-73): the forward products
// and the row log-sum-exp, about a third of the work.
template <int PARK, bool UPRIOR, int BF3, typename SH, bool GRAD = true>
__device__ float eval(KArgs& A, const float* xs, float* lw, Vec<SH::DK>& th, Vec<SH::DK>& g, bool has_temp, float temp,
                      int c, int h, int lane, bool need_value, Pace& pc, const W1Lo& wl) {
  constexpr int DKV = SH::DK;
  constexpr bool TRD = BF3 == 2 && GRAD;
  const int jj = lane & 3;
  f32x16 dW1, db1T;
  if constexpr (TRD) {
#pragma unroll
    for (int r = 0; r < 16; ++r) db1T[r] = 0.0f;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) dW1[r] = 0.0f;
  f32x4 dW0a = {0, 0, 0, 0}, dW0b = {0, 0, 0, 0}, dW2a = {0, 0, 0, 0}, dW2b = {0, 0, 0, 0};
  float db1 = 0.0f, db0 = 0.0f, db2[DKV], b2s[DKV], lik = 0.0f;
#pragma unroll
  for (int o = 0; o < DKV; ++o) {
    db2[o] = 0.0f;
    b2s[o] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, th.b2[o])));
  }
  static_assert(PARK == 0 || (PARK == 12 && DKV == 3), "all twelve elements of the three-output model or none");
  if constexpr (PARK != 0) {
    float* park = lw + WAVE_FLOATS + lane;
#pragma unroll
    for (int i = 0; i < 4; ++i) park[i * 64] = th.w0[i];
#pragma unroll
    for (int o = 0; o < 3; ++o) park[(4 + o) * 64] = th.w2[o];
    park[7 * 64] = th.b1;
    park[8 * 64] = th.b0;
#pragma unroll
    for (int o = 0; o < 3; ++o) park[(9 + o) * 64] = th.b2[o];
  }
#if EY_PHASE_TIMING
  unsigned long long ph_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const bool ph_on = (blockIdx.x & 63) == 0 && __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 0;
  unsigned long long ph_t = ph_on ? __builtin_amdgcn_s_memtime() : 0ull;
#endif
  // BF3: W1 crosses the tile loop as its three bf16 pieces (the B operand of dH0 = delta1 W1, 24 registers) instead of
  // as 16 floats; the floats are rebuilt after the loop (hi + mid + lo is exact), so they are not live inside it.
  Pieces Bw;
  if constexpr (BF3 && GRAD) {
    f32x16 w1v;
#pragma unroll
    for (int r = 0; r < 16; ++r) w1v[r] = th.w1[r];
    split16(w1v, Bw);
#pragma unroll
    for (int s = 0; s < 2; ++s) asm volatile("" : "+v"(Bw.hi[s]), "+v"(Bw.mid[s]), "+v"(Bw.lo[s]));
  }
  // One 32-row tile.  SG = the 8-row k-groups of the row-contracting products (dW2, dW1, dW0) that hold rows: 4, or 3
  // in the copy the last tile of a batch takes when its rows 24..31 are all padding (a quarter of its dW1 product).
  auto tile = [&](const int t, auto sg_tag) __attribute__((always_inline)) {
    constexpr int SG = decltype(sg_tag)::value;
    const float* xt = xs + t * XTILE_FLOATS;
    const int pace_theirs = pace_post(pc, lane);
    // ---- F0: H0^T = sigmoid(W0 X^T + b0)                                  (mlp.py:45-50)
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(lw + O_B0IMG + 8 * q + 4 * h);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[4 * q + j] = bv[j];
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(lw[O_W0IMG + c * 5 + 2 * s + h], xt[c * 5 + 2 * s + h], acc, 0, 0, 0);
    const int lab = __float_as_int(xt[c * 5 + 4]);
    const f32x16 H0 = act_tile<SH::ACT>(acc);
    if (GRAD && BF3 != 2) store_T(lw + O_TB1, H0, c, h);  // transposed copy for dW1, needed only after the backward chain: issue it early
    PH(0);
    // ---- F1: H1^T = sigmoid(W1 H0^T + b1)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(lw + O_B1IMG + 8 * q + 4 * h);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[4 * q + j] = bv[j];
    }
    if constexpr (BF3) {
      // A = the pieces of W1's row c (made once per position by write_images), B = the accumulator tile H0 itself:
      // elements 8s+j of k-step s are features 16s + 8(j>>2) + 4h + (j&3) in both
      Pieces B1p, A1p;
      split16(H0, B1p);
      if constexpr (TRD) store_pieces(lw + O_TB1, B1p, c, h);  // for dW1, read back transposed after the backward chain
      const u32x4* pv = reinterpret_cast<const u32x4*>(lw + O_W1P) + lane;
      A1p.hi[0] = pv[0 * 64]; A1p.hi[1] = pv[1 * 64];
      A1p.mid[0] = pv[2 * 64]; A1p.mid[1] = pv[3 * 64];
      if constexpr (BF3 == 2) {
        A1p.lo[0] = wl.v[0]; A1p.lo[1] = wl.v[1];
      } else {
        A1p.lo[0] = pv[4 * 64]; A1p.lo[1] = pv[5 * 64];
      }
      chain_enter(pc);
      acc = product_bf3(A1p, B1p, acc);
      chain_exit(pc);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(lw + O_W1IMG + c * TS36 + 8 * q + 4 * h);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[j], H0[4 * q + j], acc, 0, 0, 0);
      }
    }
    pace_apply(pc, pace_theirs);  // while the F1 products run
    const f32x16 H1 = act_tile<SH::ACT>(acc);
    if (GRAD) store_T(lw + O_TB0, H1, c, h);  // transposed copy for dW2; the logits and the softmax run while it lands
    PH(1);
    // ---- F2: logits = W2 H1^T + b2 with the 16-block 4x4x1 product; each half sums its 16 features
    f32x4 lg0 = {0, 0, 0, 0}, lg1 = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(lw + O_W2IMG + jj * TS36 + 8 * q + 4 * h);
      lg0 = mfma4(wv[0], H1[4 * q + 0], lg0);
      lg1 = mfma4(wv[1], H1[4 * q + 1], lg1);
      lg0 = mfma4(wv[2], H1[4 * q + 2], lg0);
      lg1 = mfma4(wv[3], H1[4 * q + 3], lg1);
    }
    float lg[DKV];
#pragma unroll
    for (int o = 0; o < DKV; ++o) lg[o] = hsum(lg0[o] + lg1[o]) + b2s[o];
    PH(2);
    float d2[DKV];
    if constexpr (SH::LIK == EY_LIK_CE_SUM) {
      // ---- CE-sum log-likelihood and output delta = onehot - softmax           (constants.py:17)
      const bool valid = lab >= 0;
      float mx = lg[0];
#pragma unroll
      for (int o = 1; o < DKV; ++o) mx = fmaxf(mx, lg[o]);
      float e[DKV], ssum = 0.0f, llab = lg[0];
#pragma unroll
      for (int o = 0; o < DKV; ++o) {
        e[o] = __expf(lg[o] - mx);
        ssum += e[o];
        if (o > 0) llab = lab == o ? lg[o] : llab;
      }
      // (need_value is wave-uniform; which lanes' rows count is a selection, not a branch: a register the allocator spills and
      // reloads inside a per-lane branch of the tile loop comes back with its other lanes lost)
      if (need_value) lik += (valid && h == 0) ? llab - (mx + __logf(ssum)) : 0.0f;
      if (!GRAD) return;  // nothing was written to LDS in this tile
      const float rs = __builtin_amdgcn_rcpf(ssum);
#pragma unroll
      for (int o = 0; o < DKV; ++o) d2[o] = valid ? ((lab == o ? 1.0f : 0.0f) - e[o] * rs) : 0.0f;
    } else {
      // ---- BCE-sum on the sigmoid output, with the naive logs of eeyore/stats/loss.py:2 (an output that rounds to 0 or
      // 1 makes the term NaN, which rejects: the reference's own f32 behaviour); the slot that holds the label of a CE
      // model holds y here, -1 marking a padding row
      const float yy = __int_as_float(lab);
      const bool valid = yy >= 0.0f;
      const float pr = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(NEG_LOG2E * lg[0]));
      if (need_value) lik += (valid && h == 0) ? __logf(pr) * yy + __logf(1.0f - pr) * (1.0f - yy) : 0.0f;
      if (!GRAD) return;
      d2[0] = valid ? (yy / pr - (1.0f - yy) / (1.0f - pr)) * (pr * (1.0f - pr)) : 0.0f;
    }
    {
      // (no per-lane branch: every lane holds its row's delta2, the lower half stores outputs 0 and 1, the upper half 2 and
      // 3 -- zeros beyond DK --, and only the lower half's copies enter the bias sums)
      const int a2 = ((c >> 2) & 1) * 16 + (c >> 3) * 4 + (c & 3);
      const bool up = h != 0;
#pragma unroll
      for (int o = 0; o < DKV; ++o) db2[o] += up ? 0.0f : d2[o];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float lo = j < DKV ? d2[j < DKV ? j : 0] : 0.0f, hi = j + 2 < DKV ? d2[j + 2 < DKV ? j + 2 : 0] : 0.0f;
        lw[O_D2BUF + (j + 2 * h) * D2S + a2] = up ? hi : lo;
      }
    }
    wave_lds_fence();
    PH(3);
    // ---- B2(2): dW2[o][k] += sum_n delta2[n][o] H1[n][k]                    (contracts over rows: transposed reads)
#pragma unroll
    for (int s = 0; s < SG; ++s) {
      const f32x4 hu = *reinterpret_cast<const f32x4*>(lw + O_TB0 + c * TS36 + 8 * s + 4 * h);
      const f32x4 du = *reinterpret_cast<const f32x4*>(lw + O_D2BUF + jj * D2S + h * 16 + 4 * s);
      dW2a = mfma4(du[0], hu[0], dW2a);
      dW2b = mfma4(du[1], hu[1], dW2b);
      dW2a = mfma4(du[2], hu[2], dW2a);
      dW2b = mfma4(du[3], hu[3], dW2b);
    }
    PH(4);
    // ---- B1(2): dH1^T = W2^T delta2^T, delta1 = dH1 * H1 (1 - H1)
    f32x16 D1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 wt = *reinterpret_cast<const f32x4*>(lw + O_W2TIMG + (8 * q + 4 * h + jj) * 4);
      f32x4 d = {0, 0, 0, 0};
#pragma unroll
      for (int o = 0; o < DKV; ++o) d = mfma4(wt[o], d2[o], d);
#pragma unroll
      for (int i = 0; i < 4; ++i) D1[4 * q + i] = d[i];
    }
    PH(5);
    D1 = times_dact<SH::ACT>(D1, H1);
    f32x16 H0U;  // H0 with lane <-> feature, register 4s+i <-> row 8s+4h+i
    if constexpr (TRD) {
      // ---- delta1 is split once: its pieces are the B operand of dH0 below and, read back transposed from the piece
      // image, the A operand of dW1; db1 is summed in the tile layout (reduced over the rows once per evaluation)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 s2 = f32x2{db1T[r], db1T[r + 1]} + f32x2{D1[r], D1[r + 1]};
        db1T[r] = s2[0];
        db1T[r + 1] = s2[1];
      }
      Pieces Ad;
      split16(D1, Ad);
      wave_lds_fence();  // the transposed reads of H1 (dW2) are done: delta1's image takes that buffer's place
      store_pieces(lw + O_TB0, Ad, c, h);
      wave_lds_fence();
      PH(6);
      // ---- B1(1): dH0^T = W1^T delta1^T with A = theta's own W1 registers (M = input feature, k = output feature) and
      // B = delta1's pieces: the accumulator is a T tile like H0, so delta0 = dH0 * H0 (1 - H0) takes H0 from registers
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
      chain_enter(pc);
      acc = product_bf3<true>(Bw, Ad, acc);
      chain_exit(pc);
      if constexpr ((EY_MF_GAPS & 1) != 0) {
        // ---- B2(1) with delta0 in its gaps.  A v_mfma_f32_32x32x16_bf16 holds the vector issue port for 8 of its 32 cycles;
        // plain vector instructions of the SAME wave that follow it in program order issue in the other 24 (a partner
        // wave's do not: profiles/r05_wave1_probe.txt), so delta0 = dH0 * act'(H0) and its transposed store -- the only
        // work of this tile that does not wait for dW1 -- go two elements behind each of the first eight products instead
        // of in front of all twelve.  Unpacked (packed f32 instructions never issue beside a bf16 MFMA); same arithmetic.
        PH(7);
        Pieces AdU, BhU;
        load_pieces_transposed(lw + O_TB0, AdU, h, lane);
        load_pieces_transposed(lw + O_TB1, BhU, h, lane);
        wave_lds_fence();  // delta0's copy goes where H0's image lies: behind the loads above
        __builtin_amdgcn_sched_barrier(0);
        static_for<12>([&](auto kk) {
          constexpr int k = decltype(kk)::value;
          dW1 = bf3_step<k, false>(AdU, BhU, dW1);
          GAP_PIN(dW1);
          if constexpr (k < 8) {
#pragma unroll
            for (int r = 2 * k; r < 2 * k + 2; ++r)
              lw[O_TB1 + (8 * (r >> 2) + 4 * h + (r & 3)) * TS36 + c] = dact_gap<SH::ACT>(acc[r], H0[r]);
          }
          __builtin_amdgcn_sched_barrier(0);
        });
        wave_lds_fence();
      } else {
      const f32x16 D0 = times_dact<SH::ACT>(acc, H0);
      PH(7);
      // ---- B2(1): dW1[out][in] += sum_n delta1[n][out] H0[n][in]: both operands from the piece images, transposed
      {
        Pieces AdU, BhU;
        load_pieces_transposed(lw + O_TB0, AdU, h, lane);
        load_pieces_transposed(lw + O_TB1, BhU, h, lane);
        chain_enter(pc);
        dW1 = product_bf3(AdU, BhU, dW1);
        chain_exit(pc);
      }
      // delta0 with lane <-> feature for the dW0 product, through H0's image (read by now)
      wave_lds_fence();
      store_T(lw + O_TB1, D0, c, h);
      wave_lds_fence();
      }
#pragma unroll
      for (int r = 4 * SG; r < 16; ++r) H0U[r] = 0.0f;
#pragma unroll
      for (int s = 0; s < SG; ++s) {
        const f32x4 du = *reinterpret_cast<const f32x4*>(lw + O_TB1 + c * TS36 + 8 * s + 4 * h);
#pragma unroll
        for (int i = 0; i < 4; ++i) H0U[4 * s + i] = du[i];
      }
    } else {
    wave_lds_fence();
    store_T(lw + O_TB0, D1, c, h);
    wave_lds_fence();
    PH(6);
    // ---- B2(1): dW1[out][in] += sum_n delta1[n][out] H0[n][in];  db1 += sum_n delta1
    f32x16 D1U;  // BF3: delta1 likewise (lane <-> output feature)
#pragma unroll
    for (int r = 4 * SG; r < 16; ++r) { H0U[r] = 0.0f; D1U[r] = 0.0f; }
#pragma unroll
    for (int s = 0; s < SG; ++s) {
      const f32x4 du = *reinterpret_cast<const f32x4*>(lw + O_TB0 + c * TS36 + 8 * s + 4 * h);
      const f32x4 hu = *reinterpret_cast<const f32x4*>(lw + O_TB1 + c * TS36 + 8 * s + 4 * h);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        H0U[4 * s + i] = hu[i];
        if constexpr (BF3) D1U[4 * s + i] = du[i];
        else if (EY_ABLATE & 8) dW1[i] += du[i] * hu[i];
        else dW1 = __builtin_amdgcn_mfma_f32_32x32x2f32(du[i], hu[i], dW1, 0, 0, 0);
      }
      db1 += (du[0] + du[1]) + (du[2] + du[3]);
    }
    if constexpr (BF3) {
      // both operands hold row 8s+4h+i as element 4s+i: any common order of the contracted index serves
      Pieces Ad, Bh;
      split16(D1U, Ad);
      split16(H0U, Bh);
      dW1 = product_bf3(Ad, Bh, dW1);
    }
    PH(7);
    // ---- B1(1): dH0 = delta1 W1 computed UNtransposed (A = delta1 tile with M = rows, B = theta's own W1
    // registers), so its accumulator is already lane <-> input feature, register <-> row: delta0 = dH0 * H0 (1 - H0)
    // comes out in the layout the dW0 product needs, with no LDS round trip (layer 0 is the last consumer).
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    if constexpr (BF3) {
      Pieces Ad;
      split16(D1, Ad);
      acc = product_bf3(Ad, Bw, acc);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(D1[r], th.w1[r], acc, 0, 0, 0);
    }
    }
    const f32x16 D0u = TRD ? H0U : times_dact<SH::ACT>(acc, H0U);  // (TRD: H0U holds delta0 itself)
    PH(8);
    // ---- B2(0): dW0[out][in] += sum_n delta0[n][out] x[n][in];  db0 += sum_n delta0
    const float* x2 = xt + 160 + jj * D2S + h * 16;
#pragma unroll
    for (int s = 0; s < SG; ++s) {
      const f32x4 du = {D0u[4 * s], D0u[4 * s + 1], D0u[4 * s + 2], D0u[4 * s + 3]};
      const f32x4 xu = *reinterpret_cast<const f32x4*>(x2 + 4 * s);
      dW0a = mfma4(du[0], xu[0], dW0a);
      dW0b = mfma4(du[1], xu[1], dW0b);
      dW0a = mfma4(du[2], xu[2], dW0a);
      dW0b = mfma4(du[3], xu[3], dW0b);
      db0 += (du[0] + du[1]) + (du[2] + du[3]);
    }
    wave_lds_fence();
    PH(9);
  };
  const int nfull = A.ntiles - A.short_last;
#pragma unroll 1
  for (int t = 0; t < nfull; ++t) tile(t, std::integral_constant<int, 4>());
  if (A.short_last) tile(nfull, std::integral_constant<int, 3>());
  if constexpr (BF3 && GRAD) {
#pragma unroll
    for (int s = 0; s < 2; ++s) asm volatile("" : "+v"(Bw.hi[s]), "+v"(Bw.mid[s]), "+v"(Bw.lo[s]));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int s = r >> 3, d = (r & 7) >> 1;
      const unsigned hh = Bw.hi[s][d], mm_ = Bw.mid[s][d], ll = Bw.lo[s][d];
      const float fh = __builtin_bit_cast(float, (r & 1) ? (hh & 0xffff0000u) : (hh << 16));
      const float fm = __builtin_bit_cast(float, (r & 1) ? (mm_ & 0xffff0000u) : (mm_ << 16));
      const float fl = __builtin_bit_cast(float, (r & 1) ? (ll & 0xffff0000u) : (ll << 16));
      th.w1[r] = (fh + fm) + fl;
    }
  }
  if constexpr (PARK != 0) {
    int at = WAVE_FLOATS + lane;
    asm volatile("" : "+v"(at));  // an offset the compiler cannot match with the stores above: no forwarding
#pragma unroll
    for (int i = 0; i < 4; ++i) th.w0[i] = lw[at + i * 64];
#pragma unroll
    for (int o = 0; o < 3; ++o) th.w2[o] = lw[at + (4 + o) * 64];
    th.b1 = lw[at + 7 * 64];
    th.b0 = lw[at + 8 * 64];
#pragma unroll
    for (int o = 0; o < 3; ++o) th.b2[o] = lw[at + (9 + o) * 64];
  }
  if constexpr (TRD) {
    // db1: the tile-layout sums (lane <-> row) through the transpose buffer, each lane then adds its feature's 16 rows
    wave_lds_fence();
    store_T(lw + O_TB0, db1T, c, h);
    wave_lds_fence();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const f32x4 du = *reinterpret_cast<const f32x4*>(lw + O_TB0 + c * TS36 + 8 * s + 4 * h);
      db1 += (du[0] + du[1]) + (du[2] + du[3]);
    }
    wave_lds_fence();
  }
  // ---- combine the two row-parity halves and the lanes
  if (GRAD) {
#pragma unroll
    for (int r = 0; r < 16; ++r) g.w1[r] = dW1[r];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      g.w0[i] = hsum(dW0a[i] + dW0b[i]);
    }
#pragma unroll
    for (int o = 0; o < DKV; ++o) {
      g.w2[o] = hsum(dW2a[o] + dW2b[o]);
      g.b2[o] = wsum(db2[o]);
    }
    g.b1 = hsum(db1);
    g.b0 = hsum(db0);
  }
  // ---- prior (bayesian_model.py:46-50): elementwise Normal(mu, sigma); temperature scales everything (:33-34,48-49)
  float qsum = 0.0f;
  if (UPRIOR || A.prior_uniform) {  // UPRIOR: the kernel variant launched only for such priors has no other path
    const float mu0 = A.mu0, iv0 = A.iv0;
    for_each_pair(th, g, c, h, lane, [&](float& tv, float& gv, int, bool counts) {
      const float d = tv - mu0;
      if (counts) qsum += d * d * iv0;
      if (GRAD) {
        float gn = gv - d * iv0;
        if (has_temp) gn *= temp;
        gv = gn;
      }
    });
  } else {
    for_each_pair(th, g, c, h, lane, [&](float& tv, float& gv, int idx, bool counts) {
      const float d = tv - A.mu[idx];
      const float iv = A.inv_var[idx];
      if (counts) qsum += d * d * iv;
      if (GRAD) {
        float gn = gv - d * iv;
        if (has_temp) gn *= temp;
        gv = gn;
      }
    });
  }
  float prior = 0.0f;
  if (need_value) {
    lik = wsum(lik);
    prior = A.prior_const - 0.5f * wsum(qsum);
  }
  if (has_temp) { lik *= temp; prior *= temp; }
#if EY_PHASE_TIMING
  if (ph_on) {
    const float keep = lik + prior + (GRAD ? g.w1[0] + g.b1 : 0.0f);  // the epilogue's results must exist before the clock is read
    if (keep == 1.2345e-30f) g.b0 += 1.0f;
  }
  PH(10);
  if (ph_on && lane == 0) {
    for (int i = 0; i < 11; ++i) atomicAdd(&g_ey_phase[i], ph_acc[i]);
    atomicAdd(&g_ey_phase[15], 1ull);
  }
#endif
  return lik + prior;
}

// ---- The pipelined evaluation (BF3 = 3; one wave per SIMD, 512 registers, four waves per CU).
// What a SIMD overlaps (DESIGN.md 4.1.2, MI355X_MICROARCH.md constants table): a v_mfma_f32_32x32x16_bf16 holds the
// vector issue port for 8 of its 32 cycles, and plain / transcendental / convert instructions that FOLLOW it in the same
// wave's program order issue in the other 24 -- but inside one row tile everything is one dependency chain (sigmoid ->
// split -> product -> sigmoid ...), and the partner wave of a SIMD cannot use the gap either (a wave whose next
// instruction is an MFMA waiting for the pipe holds the port).  So this form keeps TWO row tiles in flight in one wave:
// the backward half of tile t runs against the forward half of tile t + 1, phase by phase, every bf16 product of one
// tile issued with the vector work of the other placed in its gaps by hand (a scheduling barrier per gap):
//   ph0  F0(t+1), sigmoid, split of H0(t+1)                                      (vector work only)
//   ph1  F1(t+1): 12 MFMAs      ||  delta1(t) = dH1 * act'(H1), db1 sums, split of delta1(t)
//   ph2  dH0(t):  12 MFMAs      ||  H1(t+1) = act(F1), its transposed store
//   ph3  logits(t+1) on the 4x4x1 products; the transposed piece loads of dW1(t)
//   ph4  dW1(t):  12 MFMAs      ||  delta0(t) = dH0 * act'(H0) and its transposed store, softmax / delta2 of t+1
//   ph5  dW0(t), then dW2(t+1) and dH1(t+1) on the 4x4x1 products
// The arithmetic of a tile is eval<BF3 = 2>'s, operation for operation and in the same order of accumulation over the
// tiles (the row-contracting products take all four 8-row k-groups of a short last tile: its padding rows add zeros).
template <int ACT>
__device__ __forceinline__ float act_one(float a) {
  if (ACT == EY_ACT_RELU) return fmaxf(a, 0.0f);
  float hh = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(a) + 1.0f);
  if (ACT == EY_ACT_TANH) hh = __builtin_fmaf(2.0f, hh, -1.0f);
  return hh;
}
template <int ACT>
__device__ __forceinline__ float dact_one(float d, float hh) {
  if (ACT == EY_ACT_RELU) return hh > 0.0f ? d : 0.0f;
  const float dh = ACT == EY_ACT_TANH ? __builtin_fmaf(-hh, hh, 1.0f) : __builtin_fmaf(-hh, hh, hh);
  return d * dh;
}
#define PIPE_SB() __builtin_amdgcn_sched_barrier(0)
// An MFMA is a pure value to the instruction selector, which is free to emit it on the far side of any number of scheduling
// barriers (it sank all twelve products of a phase below the phase's last barrier).  The empty volatile statement takes
// the accumulator in and out: the MFMA that produced it cannot come later, the one that consumes it not earlier.
#define PIPE_PIN(x) asm volatile("" : "+v"(x))
template <bool UPRIOR, typename SH>
__device__ float eval_pipe(KArgs& A, const float* xs, float* lw, Vec<SH::DK>& th, Vec<SH::DK>& g, bool has_temp, float temp,
                           int c, int h, int lane, bool need_value, const W1Lo& wl) {
  constexpr int BF3 = 3;  // the LDS carve of this form (the O_* offsets)
  constexpr int DKV = SH::DK;
  constexpr int ACT = SH::ACT;
  const int jj = lane & 3;
  const bool up = h != 0;
  const int a2 = ((c >> 2) & 1) * 16 + (c >> 3) * 4 + (c & 3);
  f32x16 dW1, db1T;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dW1[r] = 0.0f; db1T[r] = 0.0f; }
  f32x4 dW0a = {0, 0, 0, 0}, dW0b = {0, 0, 0, 0}, dW2a = {0, 0, 0, 0}, dW2b = {0, 0, 0, 0};
  float db1 = 0.0f, db0 = 0.0f, db2[DKV], b2s[DKV], lik = 0.0f;
#pragma unroll
  for (int o = 0; o < DKV; ++o) {
    db2[o] = 0.0f;
    b2s[o] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, th.b2[o])));
  }
#if EY_PHASE_TIMING
  unsigned long long ph_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const bool ph_on = (blockIdx.x & 63) == 0 && __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 0;
  unsigned long long ph_t = ph_on ? __builtin_amdgcn_s_memtime() : 0ull;
#endif
  // theta's own W1 as the A operand of dH0 (its floats stay where they are: 512 registers)
  Pieces Bw;
  {
    f32x16 w1v;
#pragma unroll
    for (int r = 0; r < 16; ++r) w1v[r] = th.w1[r];
    split16(w1v, Bw);
  }
  // row c of the scaled W1, the A operand of F1, once per evaluation
  Pieces A1p;
  {
    const u32x4* pv = reinterpret_cast<const u32x4*>(lw + O_W1P) + lane;
    A1p.hi[0] = pv[0 * 64]; A1p.hi[1] = pv[1 * 64];
    A1p.mid[0] = pv[2 * 64]; A1p.mid[1] = pv[3 * 64];
    A1p.lo[0] = wl.v[0]; A1p.lo[1] = wl.v[1];
  }
  // tile t with its forward half done: H0, H1, dH1 (before the activation's factor)
  f32x16 H0c, H1c, D1c;
#pragma unroll
  for (int r = 0; r < 16; ++r) { H0c[r] = 0.0f; H1c[r] = 0.0f; D1c[r] = 0.0f; }

  auto body = [&](const int t, auto cur_tag, auto nxt_tag) __attribute__((always_inline)) {
    constexpr bool CUR = decltype(cur_tag)::value, NXT = decltype(nxt_tag)::value;
    const float* xt = xs + (t + 1) * XTILE_FLOATS;                       // the data image of tile t + 1
    const float* xc = xs + t * XTILE_FLOATS;                             // ... of tile t
    float* h0n = lw + O_TB1 + ((t + 1) & 1) * PIPE_IMG;                  // H0's piece image, by tile parity
    const float* h0c = lw + O_TB1 + (t & 1) * PIPE_IMG;
    f32x16 acc, accB, H0n, H1n, D1n;
    Pieces B1p, Ad, AdU, BhU;
    int lab = -1;
    float d2[DKV];
    f32x4 lg0 = {0, 0, 0, 0}, lg1 = {0, 0, 0, 0};
    // ---- ph0: F0, H0 = act, split, piece image of tile t + 1                   (mlp.py:45-50)
    if constexpr (NXT) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(lw + O_B0IMG + 8 * q + 4 * h);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[4 * q + j] = bv[j];
      }
#pragma unroll
      for (int s = 0; s < 2; ++s)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(lw[O_W0IMG + c * 5 + 2 * s + h], xt[c * 5 + 2 * s + h], acc, 0, 0, 0);
      lab = __float_as_int(xt[c * 5 + 4]);
#pragma unroll
      for (int r = 0; r < 16; ++r) H0n[r] = act_one<ACT>(acc[r]);
      split16(H0n, B1p);
      store_pieces(h0n, B1p, c, h);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(lw + O_B1IMG + 8 * q + 4 * h);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[4 * q + j] = bv[j];
      }
    }
    PIPE_SB();
    PH(0);
    // ---- ph1: F1(t+1) || delta1(t) = dH1 * act'(H1), db1 sums, split, piece image
    {
      f32x16 D1;
      float ra[8], rb[8];
      static_for<12>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        if constexpr (NXT) { acc = bf3_step<k, false>(A1p, B1p, acc); PIPE_PIN(acc); }
        if constexpr (CUR) {
          constexpr int j0 = (16 * k) / 12, j1 = (16 * (k + 1)) / 12;
#pragma unroll
          for (int j = j0; j < j1; ++j) {
            const int u = j >> 1, s = u >> 2, d = u & 3;
            if ((j & 1) == 0) {
              D1[2 * u] = dact_one<ACT>(D1c[2 * u], H1c[2 * u]);
              D1[2 * u + 1] = dact_one<ACT>(D1c[2 * u + 1], H1c[2 * u + 1]);
              db1T[2 * u] += D1[2 * u];
              db1T[2 * u + 1] += D1[2 * u + 1];
              Ad.hi[s][d] = split_hi(D1[2 * u], D1[2 * u + 1], ra[u], rb[u]);
            } else {
              unsigned ll;
              Ad.mid[s][d] = split_mid_lo(ra[u], rb[u], ll);
              Ad.lo[s][d] = ll;
            }
          }
        }
        PIPE_SB();
      });
      if constexpr (CUR) {
        wave_lds_fence();  // (the transposed loads of the previous tile's image were issued long ago)
        store_pieces(lw + O_TB0, Ad, c, h);
        wave_lds_fence();
      }
    }
    PIPE_SB();
    PH(1);
    // ---- ph2: dH0(t)^T = W1^T delta1^T || H1(t+1) = act(F1) and its transposed copy for dW2
    if constexpr (CUR) {
#pragma unroll
      for (int r = 0; r < 16; ++r) accB[r] = 0.0f;
    }
    static_for<12>([&](auto kk) {
      constexpr int k = decltype(kk)::value;
      if constexpr (CUR) { accB = bf3_step<k, true>(Bw, Ad, accB); PIPE_PIN(accB); }
      if constexpr (NXT) {
        // sixteen elements over gaps 1 .. 11 (the last F1 product is still running in gap 0): two each in 1 .. 5, one in 6 .. 11
        constexpr int e0 = k == 0 ? 0 : (k <= 5 ? 2 * (k - 1) : 10 + (k - 6)), e1 = k == 0 ? 0 : (k <= 5 ? 2 * k : 11 + (k - 6));
#pragma unroll
        for (int r = e0; r < e1; ++r) {
          H1n[r] = act_one<ACT>(acc[r]);
          lw[O_H1T + (8 * (r >> 2) + 4 * h + (r & 3)) * TS36 + c] = H1n[r];
        }
      }
      PIPE_SB();
    });
    PH(2);
    // ---- ph3: the transposed piece loads of dW1(t); logits(t+1) = W2 H1^T + b2 on the 4x4x1 products
    if constexpr (CUR) {
      wave_lds_fence();
      load_pieces_transposed(lw + O_TB0, AdU, h, lane);
      load_pieces_transposed(h0c, BhU, h, lane);
    }
    if constexpr (NXT) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(lw + O_W2IMG + jj * TS36 + 8 * q + 4 * h);
        lg0 = mfma4(wv[0], H1n[4 * q + 0], lg0);
        lg1 = mfma4(wv[1], H1n[4 * q + 1], lg1);
        lg0 = mfma4(wv[2], H1n[4 * q + 2], lg0);
        lg1 = mfma4(wv[3], H1n[4 * q + 3], lg1);
      }
    }
    PIPE_SB();
    PH(3);
    // ---- ph4: dW1(t) += delta1^T H0 || delta0(t) = dH0 * act'(H0) with its transposed copy; softmax / delta2 of t + 1
    {
      float lg[DKV], e[DKV], mx = 0.0f, ssum = 1.0f, llab = 0.0f, rs = 0.0f;
      bool valid = false;
      static_for<12>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        if constexpr (CUR) { dW1 = bf3_step<k, false>(AdU, BhU, dW1); PIPE_PIN(dW1); }
        if constexpr (CUR && k < 8) {
#pragma unroll
          for (int r = 2 * k; r < 2 * k + 2; ++r) {
            const float d0 = dact_one<ACT>(accB[r], H0c[r]);
            lw[O_D0T + (8 * (r >> 2) + 4 * h + (r & 3)) * TS36 + c] = d0;
          }
        }
        if constexpr (NXT && SH::LIK == EY_LIK_CE_SUM) {
          // CE-sum log-likelihood and output delta = onehot - softmax           (constants.py:17)
          if constexpr (k == 1) {
#pragma unroll
            for (int o = 0; o < DKV; ++o) lg[o] = hsum(lg0[o] + lg1[o]) + b2s[o];
          }
          if constexpr (k == 3) {
            valid = lab >= 0;
            mx = lg[0];
#pragma unroll
            for (int o = 1; o < DKV; ++o) mx = fmaxf(mx, lg[o]);
            ssum = 0.0f;
            llab = lg[0];
#pragma unroll
            for (int o = 0; o < DKV; ++o) {
              e[o] = __expf(lg[o] - mx);
              ssum += e[o];
              if (o > 0) llab = lab == o ? lg[o] : llab;
            }
          }
          if constexpr (k == 5) {
            // (need_value is wave-uniform; which lanes' rows count is a selection, not a branch: a register the allocator spills and
      // reloads inside a per-lane branch of the tile loop comes back with its other lanes lost)
      if (need_value) lik += (valid && h == 0) ? llab - (mx + __logf(ssum)) : 0.0f;
            rs = __builtin_amdgcn_rcpf(ssum);
          }
          if constexpr (k == 7) {
#pragma unroll
            for (int o = 0; o < DKV; ++o) d2[o] = valid ? ((lab == o ? 1.0f : 0.0f) - e[o] * rs) : 0.0f;
          }
        }
        if constexpr (NXT && SH::LIK != EY_LIK_CE_SUM) {
          // BCE-sum on the sigmoid output with the naive logs of eeyore/stats/loss.py:2 (see eval)
          if constexpr (k == 1) lg[0] = hsum(lg0[0] + lg1[0]) + b2s[0];
          if constexpr (k == 5) {
            const float yy = __int_as_float(lab);
            valid = yy >= 0.0f;
            const float pr = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(NEG_LOG2E * lg[0]));
            if (need_value) lik += (valid && h == 0) ? __logf(pr) * yy + __logf(1.0f - pr) * (1.0f - yy) : 0.0f;
            d2[0] = valid ? (yy / pr - (1.0f - yy) / (1.0f - pr)) * (pr * (1.0f - pr)) : 0.0f;
          }
        }
        if constexpr (NXT && k == 9) {
          // (no per-lane branch: the lower half stores outputs 0 and 1, the upper half 2 and 3 -- zeros beyond DK --, and only
          // the lower half's copies enter the bias sums)
#pragma unroll
          for (int o = 0; o < DKV; ++o) db2[o] += up ? 0.0f : d2[o];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float lo = j < DKV ? d2[j < DKV ? j : 0] : 0.0f, hi = j + 2 < DKV ? d2[j + 2 < DKV ? j + 2 : 0] : 0.0f;
            lw[O_D2BUF + (j + 2 * h) * D2S + a2] = up ? hi : lo;
          }
        }
        PIPE_SB();
      });
    }
    PH(4);
    // ---- ph5: dW0(t) += delta0^T x, db0 (delta0 with lane <-> feature through its transposed copy)
    if constexpr (CUR) {
      wave_lds_fence();
      const float* x2 = xc + 160 + jj * D2S + h * 16;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const f32x4 du = *reinterpret_cast<const f32x4*>(lw + O_D0T + c * TS36 + 8 * s + 4 * h);
        const f32x4 xu = *reinterpret_cast<const f32x4*>(x2 + 4 * s);
        dW0a = mfma4(du[0], xu[0], dW0a);
        dW0b = mfma4(du[1], xu[1], dW0b);
        dW0a = mfma4(du[2], xu[2], dW0a);
        dW0b = mfma4(du[3], xu[3], dW0b);
        db0 += (du[0] + du[1]) + (du[2] + du[3]);
      }
    }
    PIPE_SB();
    PH(5);
    // ---- dW2(t+1) += delta2^T H1 (transposed reads), dH1(t+1)^T = W2^T delta2^T
    if constexpr (NXT) {
      wave_lds_fence();
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const f32x4 hu = *reinterpret_cast<const f32x4*>(lw + O_H1T + c * TS36 + 8 * s + 4 * h);
        const f32x4 du = *reinterpret_cast<const f32x4*>(lw + O_D2BUF + jj * D2S + h * 16 + 4 * s);
        dW2a = mfma4(du[0], hu[0], dW2a);
        dW2b = mfma4(du[1], hu[1], dW2b);
        dW2a = mfma4(du[2], hu[2], dW2a);
        dW2b = mfma4(du[3], hu[3], dW2b);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 wt = *reinterpret_cast<const f32x4*>(lw + O_W2TIMG + (8 * q + 4 * h + jj) * 4);
        f32x4 d = {0, 0, 0, 0};
#pragma unroll
        for (int o = 0; o < DKV; ++o) d = mfma4(wt[o], d2[o], d);
#pragma unroll
        for (int i = 0; i < 4; ++i) D1n[4 * q + i] = d[i];
      }
      wave_lds_fence();
      H0c = H0n;
      H1c = H1n;
      D1c = D1n;
    }
    PIPE_SB();
    PH(6);
  };
  const int T = A.ntiles;
  body(-1, std::false_type(), std::true_type());
#pragma unroll 1
  for (int t = 0; t + 1 < T; ++t) body(t, std::true_type(), std::true_type());
  body(T - 1, std::true_type(), std::false_type());

  // db1: the tile-layout sums (lane <-> row) through H1's transpose buffer, each lane then adds its feature's 16 rows
  wave_lds_fence();
  store_T(lw + O_H1T, db1T, c, h);
  wave_lds_fence();
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const f32x4 du = *reinterpret_cast<const f32x4*>(lw + O_H1T + c * TS36 + 8 * s + 4 * h);
    db1 += (du[0] + du[1]) + (du[2] + du[3]);
  }
  wave_lds_fence();
  // ---- combine the two row-parity halves and the lanes
#pragma unroll
  for (int r = 0; r < 16; ++r) g.w1[r] = dW1[r];
#pragma unroll
  for (int i = 0; i < 4; ++i) g.w0[i] = hsum(dW0a[i] + dW0b[i]);
#pragma unroll
  for (int o = 0; o < DKV; ++o) {
    g.w2[o] = hsum(dW2a[o] + dW2b[o]);
    g.b2[o] = wsum(db2[o]);
  }
  g.b1 = hsum(db1);
  g.b0 = hsum(db0);
  // ---- prior (bayesian_model.py:46-50): elementwise Normal(mu, sigma); temperature scales everything (:33-34,48-49)
  float qsum = 0.0f;
  if (UPRIOR || A.prior_uniform) {
    const float mu0 = A.mu0, iv0 = A.iv0;
    for_each_pair(th, g, c, h, lane, [&](float& tv, float& gv, int, bool counts) {
      const float d = tv - mu0;
      if (counts) qsum += d * d * iv0;
      float gn = gv - d * iv0;
      if (has_temp) gn *= temp;
      gv = gn;
    });
  } else {
    for_each_pair(th, g, c, h, lane, [&](float& tv, float& gv, int idx, bool counts) {
      const float d = tv - A.mu[idx];
      const float iv = A.inv_var[idx];
      if (counts) qsum += d * d * iv;
      float gn = gv - d * iv;
      if (has_temp) gn *= temp;
      gv = gn;
    });
  }
  float prior = 0.0f;
  if (need_value) {
    lik = wsum(lik);
    prior = A.prior_const - 0.5f * wsum(qsum);
  }
  if (has_temp) { lik *= temp; prior *= temp; }
#if EY_PHASE_TIMING
  if (ph_on) {
    const float keep = lik + prior + g.w1[0] + g.b1;  // the epilogue's results must exist before the clock is read
    if (keep == 1.2345e-30f) g.b0 += 1.0f;
  }
  PH(10);
  if (ph_on && lane == 0) {
    for (int i = 0; i < 11; ++i) atomicAdd(&g_ey_phase[i], ph_acc[i]);
    atomicAdd(&g_ey_phase[15], 1ull);
  }
#endif
  return lik + prior;
}

template <int PARK, bool UPRIOR, int BF3, typename SH, bool GRAD = true>
__device__ __forceinline__ float eval_any(KArgs& A, const float* xs, float* lw, Vec<SH::DK>& th, Vec<SH::DK>& g, bool has_temp,
                                          float temp, int c, int h, int lane, bool need_value, Pace& pc, const W1Lo& wl) {
  if constexpr (BF3 == 3) {
    if constexpr (GRAD) return eval_pipe<UPRIOR, SH>(A, xs, lw, th, g, has_temp, temp, c, h, lane, need_value, wl);
    else return 0.0f;  // (the pipelined form is instantiated for modes that take the gradient)
  } else {
    return eval<PARK, UPRIOR, BF3, SH, GRAD>(A, xs, lw, th, g, has_temp, temp, c, h, lane, need_value, pc, wl);
  }
}

// Attached running moments: s1 += theta, s2 += theta^2, acc += accepted for the state this chain is left in, with the
// arithmetic of ey_stats_update (the product of two floats is exact in double).  `now` holds the state where the lane
// has it in registers; `from_memory` says to take it from theta in HBM instead (a rejected HMC proposal).
template <int DKV>
__device__ __forceinline__ void add_moments(KArgs& A, int64_t chain, Vec<DKV>& now, const float* thg, bool from_memory,
                                            bool accepted, int c, int h, int lane) {
  double* m1 = A.mom_s1 + chain * NPAR;
  double* m2 = A.mom_s2 + chain * NPAR;
  // (a pointer the compiler cannot identify with the one theta was first read through: otherwise it keeps those 29
  // values in registers through the whole trajectory instead of reading them again here)
  asm volatile("" : "+s"(thg));
  // In two halves (the 16 W1 elements, then the 13 others): inside a half every load of the read-modify-write is issued
  // before the first use (sched_barrier), so the wave waits for two memory round trips per draw, not for one per group
  // of loads -- and not for one only, which would take 145 registers at once and push the values that live across the
  // whole iteration (addresses, lane constants) into scratch, from where every leapfrog step then reloads them.
  float x[26 + DKV];
  {
    int k = 0;
    for_each(now, c, h, lane, [&](float& v, int idx, bool counts) {
      x[k] = from_memory ? thg[idx] : v;  // (from_memory is the same in every lane)
      ++k;
    });
  }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int k0 = half == 0 ? 0 : 16, k1 = half == 0 ? 16 : 26 + DKV;
    double a1[16], a2[16];
    int k = 0;
    for_each(now, c, h, lane, [&](float&, int idx, bool) {
      if (k >= k0 && k < k1) {
        a1[k - k0] = m1[idx];
        a2[k - k0] = m2[idx];
      }
      ++k;
    });
    __builtin_amdgcn_sched_barrier(0);
    k = 0;
    for_each(now, c, h, lane, [&](float&, int idx, bool) {
      if (k >= k0 && k < k1) {
        const double t = (double)x[k];
        m1[idx] = a1[k - k0] + t;
        m2[idx] = a2[k - k0] + t * t;
      }
      ++k;
    });
    __builtin_amdgcn_sched_barrier(0);
  }
  A.mom_acc[chain] += accepted ? 1.0 : 0.0;
}

// The chain's N(0,1) stream for all NPAR elements, generated by the wave together: lane l computes the blocks of four
// l, l + 64, ... (ey_rng_normal4: one Philox call per block) into the two transpose buffers of this wave's LDS region
// (free between evaluations), from where every lane then picks the 29 elements of its register layout.  One call
// per element in every lane, as a lane-local draw needs, costs 29 Philox calls per lane instead of at most 6.
template <int BF3, int DKV>
__device__ __forceinline__ const float* stage_normals(float* lw, const EyRng& rn, int lane) {
  float* st = lw + O_STAGE;  // the two transpose buffers are adjacent: 2304 floats >= NPAR + 3
  constexpr int NB = (NPAR + 3) / 4;
  for (int b = lane; b < NB; b += 64) {
    float o[4];
    ey_rng_normal4<float>(rn, (uint32_t)b, o);
    *reinterpret_cast<f32x4*>(st + 4 * b) = f32x4{o[0], o[1], o[2], o[3]};
  }
  wave_lds_fence();
  return st;
}

// One chain of one launch: everything between reading theta and writing the accepted state back.
template <int MODE, int PARK, bool UPRIOR, bool DA, int BF3, typename SH>
__device__ __forceinline__ void run_chain(KArgs& A, const float* xs, float* lw, const int64_t chain, const int it,
                                          const int c, const int h, const int lane, Pace& pc) {
  constexpr int DKV = SH::DK;
  typedef Vec<DKV> Vec;
  // Later iterations of one launch read what this wave's lanes wrote at the end of the previous one.  Workgroup scope
  // is enough (and only costs a wait for the outstanding stores): all accesses go through this CU's vector cache in
  // order; an agent-scope release would write the whole L2 of the XCD back every iteration.
  if (it > 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const uint64_t iter = A.iter + (uint64_t)it;
#if EY_PHASE_TIMING
  const bool kt_on = (blockIdx.x & 63) == 0 && pc.wave == 0;
  if (lane == 0 && chain < 8192) g_ey_wave_t[3 * chain + 1] = __builtin_amdgcn_s_memrealtime();
  const unsigned long long kt0 = kt_on ? __builtin_amdgcn_s_memtime() : 0ull;
  unsigned long long ko_acc[6] = {0, 0, 0, 0, 0, 0}, ko_t = kt0;
#endif
  float* thg = A.theta + chain * NPAR;
  float* grg = A.grad + chain * NPAR;
  const bool has_temp = A.temp != nullptr;
  const float temp = has_temp ? A.temp[chain] : 1.0f;
  const float eps = A.step_vec ? A.step_vec[chain] : A.step;

  Vec th, g, p;
  W1Lo wl;
  for_each(th, c, h, lane, [&](float& v, int idx, bool) { v = thg[idx]; });

  if (MODE == MODE_GRAD) {
    write_images<BF3, SH>(lw, th, c, h, lane, wl);
    const float t = eval_any<PARK, UPRIOR, BF3, SH>(A, xs, lw, th, g, has_temp, temp, c, h, lane, true, pc, wl);
    for_each(g, c, h, lane, [&](float& v, int idx, bool counts) { grg[idx] = v; });
    A.target[chain] = t;
    return;
  }

  if (MODE == MODE_MALA || MODE == MODE_MH) {
    // MALA.draw (eeyore/samplers/mala.py:46-82) / MetropolisHastings.draw (metropolis_hastings.py:41-73): one
    // evaluation at the proposal; p holds the proposal, gp its gradient
    const EyRng rn = ey_rng_make(A.seed, A.chain_offset + (uint64_t)chain, iter, EY_STREAM_NORMAL);
    const float* zin = A.p0 ? A.p0 + chain * NPAR : nullptr;
    const float sc = A.step_vec ? sqrtf(eps) : A.sqrt_step;  // scale = sqrt(step), mala.py:39
    float qf = 0.0f;
    Vec gp;
    if (MODE == MODE_MALA) for_each(g, c, h, lane, [&](float& v, int idx, bool) { v = grg[idx]; });
    const float* zst = zin ? nullptr : stage_normals<BF3, DKV>(lw, rn, lane);
    for_each3(th, g, p, c, h, lane, [&](float& tv, float& gv, float& pv, int idx, bool counts) {
      const float zi = zin ? zin[idx] : zst[idx];
      if (MODE == MODE_MALA) {
        const float loc = tv + 0.5f * eps * gv;  // kernel_mean, mala.py:35-36
        pv = loc + sc * zi;
        const float d = pv - loc;
        if (counts) qf += d * d;
      } else {
        pv = tv + A.scale[idx] * zi;  // NormalKernel(theta, scale).sample()
      }
    });
    wave_lds_fence();  // the staged normals have been read; the evaluation reuses that LDS
    write_images<BF3, SH>(lw, p, c, h, lane, wl);
    const float tv = eval_any<PARK, UPRIOR, BF3, SH, MODE == MODE_MALA>(A, xs, lw, p, gp, has_temp, temp, c, h, lane, true, pc, wl);
    const float t_old = A.target[chain];
    float log_rate = tv - t_old;  // symmetric kernel: metropolis_hastings.py:50
    if (MODE == MODE_MALA) {
      float qb = 0.0f;
      for_each3(th, gp, p, c, h, lane, [&](float& tv0, float& gpv, float& pv, int, bool counts) {
        const float d = tv0 - (pv + 0.5f * eps * gpv);
        if (counts) qb += d * d;
      });
      const float inv2v = 1.0f / (2.0f * sc * sc);
      log_rate += (wsum(qf) - wsum(qb)) * inv2v;  // the -P log s - P/2 log 2pi terms cancel (mala.py:58-64)
    }
    const EyRng ru = ey_rng_make(A.seed, A.chain_offset + (uint64_t)chain, iter, EY_STREAM_UNIFORM);
    const float u = A.u ? A.u[chain] : ey_rng_uniform<float>(ru);
    const bool acc = __builtin_amdgcn_readfirstlane((int)(__logf(u) < log_rate)) != 0;  // mala.py:66, metropolis_hastings.py:56 (a scalar)
    if (acc) {
      for_each(p, c, h, lane, [&](float& v, int idx, bool counts) { thg[idx] = v; });
      if (MODE == MODE_MALA)
        for_each(gp, c, h, lane, [&](float& v, int idx, bool counts) { grg[idx] = v; });
    }
    {  // (every lane stores the same value: no store behind a per-lane branch)
      if (acc) A.target[chain] = tv;
      A.accepted[chain] = acc ? 1 : 0;
      if (A.rate) A.rate[chain] = log_rate;
      if (A.rec_targets) A.rec_targets[(int64_t)it * A.C + chain] = acc ? tv : t_old;
      if (A.rec_accepted) A.rec_accepted[(int64_t)it * A.C + chain] = acc ? 1 : 0;
      if (A.accept_count && acc) A.accept_count[chain] += 1;
    }
    if (A.mom_s1) add_moments(A, chain, acc ? p : th, thg, false, acc, c, h, lane);
    if (A.rec_samples) {  // the state this chain is left in (chain_list.py:64-67): both candidates are in registers
      float* so = A.rec_samples + ((int64_t)it * A.C + chain) * NPAR;
      for_each2(th, p, [&](float& tv0, float& pv) { tv0 = acc ? pv : tv0; });
      for_each(th, c, h, lane, [&](float& v, int idx, bool counts) { so[idx] = v; });
    }
    if (A.n_iters > 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    return;
  }

  float t_cur = 0.0f, kin = 0.0f;
  if (MODE == MODE_HMC) {
    const EyRng rn = ey_rng_make(A.seed, A.chain_offset + (uint64_t)chain, iter, EY_STREAM_NORMAL);
    const float* p0 = A.p0 ? A.p0 + chain * NPAR : nullptr;
    const float* pst = p0 ? nullptr : stage_normals<BF3, DKV>(lw, rn, lane);
    for_each(p, c, h, lane, [&](float& v, int idx, bool counts) {
      v = p0 ? p0[idx] : pst[idx];   // hmc.py:134
      if (counts) kin += v * v;
    });
    wave_lds_fence();  // the staged normals have been read; the evaluations reuse that LDS
    kin = wsum(kin);
    t_cur = A.target[chain];
  } else {
    const float* pin = A.pio + chain * NPAR;
    for_each(p, c, h, lane, [&](float& v, int idx, bool) { v = pin[idx]; });
  }
  const float h_cur = -t_cur + 0.5f * kin;  // hmc.py:91-98,137
  float t = t_cur;
  // leapfrog, hmc.py:100-124 (grad_potential = -grad): the first half step of the momentum
  if (MODE == MODE_LEAPFROG || A.recompute) {  // hmc.py:104
    write_images<BF3, SH>(lw, th, c, h, lane, wl);
    t = eval_any<PARK, UPRIOR, BF3, SH>(A, xs, lw, th, g, has_temp, temp, c, h, lane, true, pc, wl);
    for_each2(p, g, [&](float& pv, float& gv) { pv = __builtin_fmaf(0.5f * eps, gv, pv); });
  } else {
    // the cached gradient goes from memory straight into the momentum: loaded into a register vector of its own before
    // the momentum existed it was spilled pair by pair, every pair behind its own s_waitcnt vmcnt(0) (15 serial round
    // trips per draw)
    // (one fused multiply-add per element, written out: left to the compiler the two forms of this update were contracted
    // differently -- v_fmac_f32 there, v_pk_mul_f32 + v_pk_add_f32 here -- and a chain's bits depended on the flag)
    for_each(p, c, h, lane, [&](float& pv, int idx, bool) { pv = __builtin_fmaf(0.5f * eps, grg[idx], pv); });
  }
#pragma unroll 1
  KO(0);
  for (int k = 1; k <= A.L; ++k) {
    for_each2(th, p, [&](float& tv, float& pv) { tv = tv + eps * pv; });
    KO(1);
    write_images<BF3, SH>(lw, th, c, h, lane, wl);
    KO(2);
    t = eval_any<PARK, UPRIOR, BF3, SH>(A, xs, lw, th, g, has_temp, temp, c, h, lane, k == A.L, pc, wl);
    KO(3);
    const float w = (k < A.L) ? eps : 0.5f * eps;
    for_each2(p, g, [&](float& pv, float& gv) { pv = pv + w * gv; });
    KO(4);
  }

  if (MODE == MODE_LEAPFROG) {
    float* pout = A.pio + chain * NPAR;
    for_each(th, c, h, lane, [&](float& v, int idx, bool counts) { thg[idx] = v; });
    for_each(p, c, h, lane, [&](float& v, int idx, bool counts) { pout[idx] = -v; });  // hmc.py:122
    for_each(g, c, h, lane, [&](float& v, int idx, bool counts) { grg[idx] = v; });
    A.target[chain] = t;
    return;
  }

  // the pointers the write-back uses are made opaque here: otherwise the per-lane addresses theta + idx, grad + idx are
  // computed before the trajectory, live across it and spilled (14 dwords per lane stored and reloaded per draw)
  asm volatile("" : "+s"(thg), "+s"(grg));
  // ... and so are the lane coordinates everything below indexes with: the element offsets of the write-back, the moments and
  // the record were otherwise computed at kernel entry (they depend on the lane alone), kept for the whole launch and spilled
  // (22 dwords per lane written at entry and read back in every draw)
  int le = lane;
  asm volatile("" : "+v"(le));
  const int ce = le & 31, he = le >> 5;
  kin = 0.0f;
  for_each(p, ce, he, le, [&](float& v, int, bool counts) { if (counts) kin += v * v; });
  kin = wsum(kin);
  const float h_prop = -t + 0.5f * kin;
  float rate = __expf(h_cur - h_prop);  // hmc.py:143-146
  if (rate > 1.0f) rate = 1.0f;
  const EyRng ru = ey_rng_make(A.seed, A.chain_offset + (uint64_t)chain, iter, EY_STREAM_UNIFORM);
  const float u = A.u ? A.u[chain] : ey_rng_uniform<float>(ru);
  // strict <, NaN => reject (hmc.py:148); the same in every lane, and read as a scalar so that the stores behind it are behind
  // a scalar branch
  const bool acc = __builtin_amdgcn_readfirstlane((int)(u < rate)) != 0;
  if (acc) {
    for_each(th, ce, he, le, [&](float& v, int idx, bool counts) { thg[idx] = v; });
    for_each(g, ce, he, le, [&](float& v, int idx, bool counts) { grg[idx] = v; });
  }
  {  // (every lane stores the same value: no store behind a per-lane branch)
    if (acc) A.target[chain] = t;
    A.accepted[chain] = acc ? 1 : 0;
    if (A.rate) A.rate[chain] = rate;
    if (A.hcur) A.hcur[chain] = h_cur;
    if (A.hprop) A.hprop[chain] = h_prop;
    // the tuner step of hmc.py:158-163, per chain, without leaving the launch.  A template flag: in the instantiation the
    // headline benchmark runs (no tuner attached) this code is absent -- its f64 polynomial constants were hoisted to the
    // top of the kernel, spilled, and reloaded in every draw, and without it the register allocation of the whole
    // kernel comes out 2 % faster.  A launch with a tuner attached takes the twin instantiation with the same arithmetic.
    if constexpr (DA) {
      if (A.da_state && it < A.da_n)
        A.da_step[chain] = (float)ey_da_update(A.da_state + 3 * chain, A.da_tab + 3 * it, (double)rate, A.da_d,
                                               A.da_has_eub != 0, A.da_logeub, it == A.da_final_it);
    }
  }
  if (A.mom_s1) add_moments(A, chain, th, thg, !acc, acc, ce, he, le);
  if (A.rec_samples) {  // the state this chain is left in (what ChainList.update stores, chain_list.py:64-67)
    float* so = A.rec_samples + ((int64_t)it * A.C + chain) * NPAR;
    const float* old = thg;
    asm volatile("" : "+s"(old));  // see add_moments
    for_each(th, ce, he, le, [&](float& v, int idx, bool counts) { so[idx] = acc ? v : old[idx]; });
  }
  {  // (every lane stores the same value: no store behind a per-lane branch)
    if (A.rec_targets) A.rec_targets[(int64_t)it * A.C + chain] = acc ? t : t_cur;
    if (A.rec_accepted) A.rec_accepted[(int64_t)it * A.C + chain] = acc ? 1 : 0;
    if (A.accept_count && acc) A.accept_count[chain] += 1;
  }
  if (A.n_iters > 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
#if EY_PHASE_TIMING
  if (lane == 0 && chain < 8192) g_ey_wave_t[3 * chain + 2] = __builtin_amdgcn_s_memrealtime();
  KO(5);
  if (kt_on && lane == 0) {
    for (int i = 0; i < 6; ++i) atomicAdd(&g_ey_phase[16 + i], ko_acc[i]);
    atomicAdd(&g_ey_phase[11], __builtin_amdgcn_s_memtime() - kt0);
    atomicAdd(&g_ey_phase[14], 1ull);
  }
#endif
}

// Persistent launch: one 8-wave workgroup per CU (two waves per SIMD), every wave walks over chains
// blockIdx + gridDim*wave, + 8*gridDim, ...  so a wave starts its next chain the moment it finishes one (no workgroup
// re-dispatch between rounds, the data image is staged once) and a partial last round spreads one wave per SIMD.
// Two waves share a SIMD because f32 MFMA runs on the vector ALUs: the partner hides latency (LDS round trips, MFMA
// result latency) rather than adding throughput (tools/coexec_probe*.hip).  WAVES = 4 is the former layout (two
// 4-wave workgroups per CU, one chain per wave), kept for A/B runs (ey_debug_set_variant bit 0).
template <int MODE, int WAVES, int PARK, bool UPRIOR, bool DA, int BF3, typename SH>
__global__ void __launch_bounds__(WAVES * 64, 2) k_mfma32(MfArgs A) {
  constexpr int MF_WAVES = WAVES, MF_THREADS = WAVES * 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: chain index and addresses stay scalar
  const int c = lane & 31, h = lane >> 5;
#if EY_PHASE_TIMING
  const unsigned long long rt_entry = __builtin_amdgcn_s_memrealtime();
#endif
  // shared, read-only data images
  const int xfloats = A.ntiles * XTILE_FLOATS;
  for (int i = tid; i < xfloats; i += MF_THREADS) smem[i] = A.xpack[i];
  const float* xs = smem;
  constexpr int REGION = WAVE_FLOATS + 64 * PARK;
  float* lw = smem + xfloats + wave * REGION;
  int* ctl = reinterpret_cast<int*>(smem + xfloats + MF_WAVES * REGION);  // [waves] counters, [waves] SIMD ids
  // HW_REG_HW_ID (id 4), SIMD_ID = bits 5:4
  const int simd = (int)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);
  if (lane == 0) {
    ctl[wave] = 0;
    ctl[MF_WAVES + wave] = simd;
  }
  __syncthreads();
  Pace pc;
  pc.prog = ctl;
  pc.junk = ctl + MF_WAVES + wave;
  pc.wave = wave;
  pc.partner = wave;
  pc.left = 0;
  pc.bias = 0;
  pc.hi = false;
  if ((EY_PHASE_PRIO == 1 || EY_PHASE_PRIO == 2) && BF3 == 2) __builtin_amdgcn_s_setprio(2);  // the level of a wave outside a chain of bf16 products
  int mates = 0;
  for (int w = 0; w < MF_WAVES; ++w)
    if (w != wave && ctl[MF_WAVES + w] == simd) { pc.partner = w; ++mates; }
  pc.on = A.balance != 0 && mates == 1;
  if (pc.on && wave > pc.partner) pc.bias = A.stagger;
#if EY_PHASE_TIMING
  if (blockIdx.x == 0 && lane == 0) {
    g_ey_dbg[wave] = simd;
    g_ey_dbg[8 + wave] = pc.partner;
    g_ey_dbg[16 + wave] = mates;
    g_ey_dbg[24 + wave] = (int)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
  }
#endif
  const int64_t first = (int64_t)blockIdx.x + (int64_t)gridDim.x * wave, stride = (int64_t)gridDim.x * MF_WAVES;
  {  // row tiles this wave evaluates in the whole launch
    const int64_t mine = first < A.C ? (A.C - first + stride - 1) / stride : 0;
    const int evals = (MODE == MODE_HMC) ? A.L + (A.recompute ? 1 : 0) : (MODE == MODE_LEAPFROG ? A.L + 1 : 1);
    const int iters = (MODE == MODE_HMC || MODE == MODE_MALA || MODE == MODE_MH) ? A.n_iters : 1;
    pc.left = (int)std::min<int64_t>(mine * evals * A.ntiles * iters, 0x3fffffff);
    if (lane == 0) ctl[wave] = pc.left + pc.bias;
  }
  __syncthreads();
  if (__builtin_amdgcn_readfirstlane(ctl[pc.partner]) - (pc.partner > wave ? A.stagger : 0) == 0)
    pc.on = false;  // the partner has no chain at all
  const int n_iters = (MODE == MODE_HMC || MODE == MODE_MALA || MODE == MODE_MH) ? A.n_iters : 1;
  // Chain-major: a wave takes one of its chains through all iterations of the launch before it starts the next (the
  // chains are independent), so the 21 KB of f64 moment accumulators and the state it reads back stay in this CU's
  // caches between iterations instead of being streamed from HBM once per iteration.
  for (int64_t chain = first; chain < A.C; chain += stride) {  // whole waves; no workgroup synchronisation below
    for (int it = 0; it < n_iters; ++it) {
      // Re-read the arguments from the kernarg segment in every round: hoisted out of this loop they would all stay
      // live in scalar registers for the whole kernel (106 SGPRs, spilled into vector registers, which then spill too).
      KArgs* Ap = (KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(Ap));
#if EY_PHASE_TIMING
      if (lane == 0 && chain < 8192) g_ey_wave_t[3 * chain] = rt_entry;
#endif
      run_chain<MODE, PARK, UPRIOR, DA, BF3, SH>(*Ap, xs, lw, chain, it, c, h, lane, pc);
    }
  }
  if (lane == 0) __hip_atomic_store(&ctl[wave], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // nothing left
}

// The pipelined form's launch: one 4-wave workgroup per CU, ONE wave per SIMD (512 registers per lane), every wave walks
// over chains blockIdx + gridDim*wave, + 4*gridDim, ... chain-major like k_mfma32.  No pacing: a wave has its SIMD to itself.
template <int MODE, bool UPRIOR, bool DA, typename SH>
__global__ void __launch_bounds__(256, 1) k_mfma32p(MfArgs A) {
  constexpr int BF3 = 3;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5;
  const int xfloats = A.ntiles * XTILE_FLOATS;
  for (int i = tid; i < xfloats; i += 256) smem[i] = A.xpack[i];
  const float* xs = smem;
  float* lw = smem + xfloats + wave * WAVE_FLOATS;
  __syncthreads();
  Pace pc;
  pc.prog = nullptr; pc.junk = nullptr; pc.wave = wave; pc.partner = wave; pc.left = 0; pc.bias = 0; pc.on = false; pc.hi = false;
  const int64_t first = (int64_t)blockIdx.x + (int64_t)gridDim.x * wave, stride = (int64_t)gridDim.x * 4;
  const int n_iters = A.n_iters;
  for (int64_t chain = first; chain < A.C; chain += stride) {
    for (int it = 0; it < n_iters; ++it) {
      KArgs* Ap = (KArgs*)__builtin_amdgcn_kernarg_segment_ptr();  // (see k_mfma32)
      asm volatile("" : "+s"(Ap));
      run_chain<MODE, 0, UPRIOR, DA, BF3, SH>(*Ap, xs, lw, chain, it, c, h, lane, pc);
    }
  }
}

#ifndef EY_MF_PART
#define EY_MF_PART 0
#endif
#define MF_PIPE_TILES 8   // the pipelined form's per-wave regions (4 x 34.3 KB) leave room for 8 row tiles (N <= 256)
// (defined in the translation unit of its own, ey_mfma32p.hip = this file with EY_MF_PART 1: built without packed f32
// instructions, which never issue in the shadow of a bf16 MFMA)
int ey_mfma32p_hmc(MfArgs& a, int n_cu, hipStream_t s);
#if EY_MF_PART == 1
int ey_mfma32p_hmc(MfArgs& a, int n_cu, hipStream_t s) {
  const size_t bytes = sizeof(float) * ((size_t)a.ntiles * XTILE_FLOATS + 4 * (size_t)WAVE_FLOATS_OF(3));
  const int most = (int)(sizeof(float) * ((size_t)MF_PIPE_TILES * XTILE_FLOATS + 4 * (size_t)WAVE_FLOATS_OF(3)));
  const unsigned grid = (unsigned)std::min<int64_t>(a.C, n_cu > 0 ? n_cu : 256);
#define MFP_LAUNCH(DA)                                                                                              \
  do {                                                                                                             \
    EY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mfma32p<MODE_HMC, true, DA, MfHeadline>),            \
                               hipFuncAttributeMaxDynamicSharedMemorySize, most));                                 \
    hipLaunchKernelGGL((k_mfma32p<MODE_HMC, true, DA, MfHeadline>), dim3(grid), dim3(256), bytes, s, a);            \
  } while (0)
  if (a.da_state) MFP_LAUNCH(true);
  else MFP_LAUNCH(false);
#undef MFP_LAUNCH
  EY_HIP(hipGetLastError());
  return EY_OK;
}
#if EY_PHASE_TIMING
extern "C" int ey_debug_phase_read_p(unsigned long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_ey_phase), 32 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[32] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_ey_phase), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
#endif
#else
// ----------------------------------------------------------------------------------------------- host side
// 1: the headline model (both product forms); 2: a 4-32-32 model with another hidden activation or the BCE head, served
// in the bf16x3 form only (the exact form of those is fused16's); 0: not this kernel's
int ey_mfma32_kind(const ey_plan* pl) {
  const EyModel& m = pl->m;
  if (pl->dtype != EY_F32 || m.nl != 3) return 0;
  if (m.dims[0] != 4 || m.dims[1] != 32 || m.dims[2] != 32) return 0;
  if (!m.bias[0] || !m.bias[1] || !m.bias[2]) return 0;
  if (m.act[0] != m.act[1] || m.act[0] < EY_ACT_SIGMOID || m.act[0] > EY_ACT_RELU) return 0;
  if (m.lik == EY_LIK_CE_SUM && m.dims[3] == 3 && m.act[2] == EY_ACT_NONE) return m.act[0] == EY_ACT_SIGMOID ? 1 : 2;
  if (m.lik == EY_LIK_BCE_SUM && m.dims[3] == 1 && m.act[2] == EY_ACT_SIGMOID) return 2;
  return 0;
}
bool ey_mfma32_supports(const ey_plan* pl) { return ey_mfma32_kind(pl) != 0; }

static size_t mf_lds_bytes(int ntiles, int waves, int park, int bf3) {
  return sizeof(float) * ((size_t)ntiles * XTILE_FLOATS + (size_t)waves * (WAVE_FLOATS_OF(bf3) + 64 * park)) +
         sizeof(int) * 2 * waves;
}
#define MF_BF3_TILES 16   // the bf16x3 form's larger per-wave region leaves room for 16 row tiles (N <= 512)
#define MF_TRD_TILES 10   // ... and with the piece images of H0 and delta1 (BF3 = 2) for 10 (N <= 320)
#define MF_PARK 12        // position elements parked in LDS during the tile loop ...
#define MF_PARK_TILES 6   // ... when the data image leaves room for it (N <= 192 rows)

// kernel variant (A/B knob): bit 0 = former launch shape (4-wave workgroups, one chain per wave), bit 1 = no priority
// balancing between the two waves of a SIMD, bit 2 = no momentum parking (exact form) / no piece
// images (bf16x3 form: BF3 = 1 where BF3 = 2 would serve), bit 4 = route f32 plans through the layerwise
// path, bit 5 = the layerwise path's register-staged GEMM instead of the LDS-DMA one, bit 6 = the layerwise path's
// narrow last layer as separate launches instead of the fused tail kernel, bit 7 = the layerwise path's leapfrog update
// as a separate kernel instead of in the gradient kernels' epilogues
// ey_debug_set_variant (ey_api.hip) bits 0..3: launch variants of this kernel, for A/B runs and tests

// Pack (x, labels) into the per-tile LDS images, on the device and on the caller's stream: one thread per (tile, row).
// Rows beyond N are zero with label -1 (they contribute nothing).
__global__ void k_mf_pack(const float* __restrict__ x, const int* __restrict__ labels, const float* __restrict__ y1,
                          int N, int ntiles, float* __restrict__ img) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ntiles * 32) return;
  const int t = i >> 5, cc = i & 31, n = i;
  float* xt = img + (size_t)t * XTILE_FLOATS;
  int lab = -1;
  float xv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (n < N) {
    lab = labels[n];
#pragma unroll
    for (int k = 0; k < 4; ++k) xv[k] = x[(size_t)n * 4 + k];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) xt[cc * 5 + k] = xv[k];
  // the label of a CE model; for the BCE head (y1 given: one output) the target y itself, -1 marking a padding row
  xt[cc * 5 + 4] = y1 ? (n < N ? y1[n] : -1.0f) : __int_as_float(lab);
  // regrouped copy for the 4x4x1 weight-gradient product: [in][half][s'][i], row = 8s' + 4 half + i
  const int sp = cc >> 3, hh = (cc >> 2) & 1, ii = cc & 3;
#pragma unroll
  for (int k = 0; k < 4; ++k) xt[160 + k * D2S + hh * 16 + sp * 4 + ii] = xv[k];
  // the four pad floats of each input's row of the regrouped image are never read
}

// Asynchronous: no host round trip and no allocation per batch (the image buffer is sized once, for MF_MAX_TILES).
// A batch with more row tiles than the kernel's LDS image can hold sends this plan to the other kernel families
// until a batch that fits arrives (mfma32_data_ok is recomputed on every call).
int ey_mfma32_set_data(ey_plan* pl, hipStream_t s) {
  const EyModel& m = pl->m;
  const int ntiles = (m.N + 31) / 32;
  pl->mfma32_data_ok = ntiles <= MF_MAX_TILES;
  if (!pl->mfma32_data_ok) return EY_OK;
  if (!pl->d_xpack) EY_HIP(hipMalloc(&pl->d_xpack, sizeof(float) * (size_t)MF_MAX_TILES * XTILE_FLOATS));
  hipLaunchKernelGGL(k_mf_pack, dim3((ntiles * 32 + 255) / 256), dim3(256), 0, s, (const float*)pl->d_x,
                     (const int*)pl->d_labels, m.lik == EY_LIK_BCE_SUM ? (const float*)pl->d_y : nullptr, m.N, ntiles,
                     (float*)pl->d_xpack);
  EY_HIP(hipGetLastError());
  return EY_OK;
}

template <int MODE, int WAVES, int PARK, bool UPRIOR = false, bool DA = true, int BF3 = 0, typename SH = MfHeadline>
static int mf_launch_v(MfArgs& a, int n_cu, hipStream_t s) {
  const size_t bytes = mf_lds_bytes(a.ntiles, WAVES, PARK, BF3);
  // per launch: function attributes are per device and plans on different devices / threads share this code
  EY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mfma32<MODE, WAVES, PARK, UPRIOR, DA, BF3, SH>),
                             hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)mf_lds_bytes(BF3 == 2 ? MF_TRD_TILES : (BF3 ? MF_BF3_TILES : (PARK ? MF_PARK_TILES : MF_MAX_TILES)), WAVES, PARK, BF3)));
  // 8 waves: one persistent workgroup per CU, or one per chain when there are fewer chains than CUs (then only wave
  // 0 of a workgroup has work and every chain gets a CU to itself); 4 waves: a workgroup per 4 chains
  const unsigned grid = WAVES == 8 ? (unsigned)std::min<int64_t>(a.C, n_cu > 0 ? n_cu : 256)
                                   : (unsigned)((a.C + WAVES - 1) / WAVES);
  hipLaunchKernelGGL((k_mfma32<MODE, WAVES, PARK, UPRIOR, DA, BF3, SH>), dim3(grid), dim3(WAVES * 64), bytes, s, a);
  EY_HIP(hipGetLastError());
  return EY_OK;
}

static bool mf_moments_from_records(const ey_plan* pl, const MfArgs& a) {
  static const bool in_kernel = [] { const char* e = getenv("EY_MF_MOMENTS_IN_KERNEL"); return e && atoi(e) != 0; }();  // A/B knob
  return pl->mom_s1 && a.rec_samples && a.rec_accepted && a.n_iters > 1 && !in_kernel;
}
static int mf_finish(ey_plan* pl, const MfArgs& a, int rc, hipStream_t s) {
  if (rc != EY_OK || !mf_moments_from_records(pl, a)) return rc;
  return ey_stats_update_run(a.rec_samples, a.rec_accepted, a.n_iters, a.C, pl->m.P, pl->dtype, pl->mom_s1, pl->mom_s2,
                             pl->mom_acc, s);
}

#define MF_COMMA ,
template <int MODE>
static int mf_launch(ey_plan* pl, MfArgs& a, hipStream_t s) {
  const EyModel& m = pl->m;
  a.xpack = (const float*)pl->d_xpack;
  a.mu = (const float*)m.mu;
  a.inv_var = (const float*)m.inv_var;
  a.prior_uniform = pl->prior_uniform ? 1 : 0;
  a.mu0 = (float)pl->prior_mu0;
  a.iv0 = (float)pl->prior_iv0;
  a.prior_const = (float)m.prior_const;
  a.ntiles = (m.N + 31) / 32;
  a.short_last = (m.N - 32 * (a.ntiles - 1)) <= 24 ? 1 : 0;
  a.balance = (t_ey_variant & 2) ? 0 : 1;
  {
    // two row tiles: +0.5-0.9 % over none in three interleaved A/B rounds (profiles/r03_ab_stagger.txt); EY_MF_STAGGER overrides
    static const int stagger = [] { const char* e = getenv("EY_MF_STAGGER"); return e ? atoi(e) : 2; }();
    a.stagger = stagger;
  }
  // attached moments: accumulated by the kernel draw by draw -- unless the launch records samples and accept flags anyway
  // (mf_moments_from_records): then one streaming pass over the records behind the launch makes the same sums, bit for bit
  // (mf_finish), and the f64 accumulators are read and written once per launch instead of once per iteration
  if ((MODE == MODE_HMC || MODE == MODE_MALA || MODE == MODE_MH) && !mf_moments_from_records(pl, a)) {
    a.mom_s1 = pl->mom_s1;
    a.mom_s2 = pl->mom_s2;
    a.mom_acc = pl->mom_acc;
  }
  const bool bf3 = pl->products == EY_PRODUCTS_BF16X3 && a.ntiles <= MF_BF3_TILES && !(t_ey_variant & 1);
  if (ey_mfma32_kind(pl) == 2) {
    // the other 4-32-32 models: one instantiation per mode (bf16x3 form, any prior, in-kernel tuner compiled in)
    if (!bf3) EY_FAIL(EY_ERR_UNSUPPORTED, "mfma32: this model is served in the bf16x3 form only");
    const int act = m.act[0];
#define MF_SHAPE_LAUNCH(SHAPE)                                                                              \
  do {                                                                                                     \
    if (MODE == MODE_HMC && a.ntiles <= MF_TRD_TILES && !(t_ey_variant & 4))                               \
      return mf_launch_v<MODE, 8, 0, false, true, MODE == MODE_HMC ? 2 : 1, SHAPE>(a, pl->n_cu, s);       \
    return mf_launch_v<MODE, 8, 0, false, true, 1, SHAPE>(a, pl->n_cu, s);                                 \
  } while (0)
    if (m.lik == EY_LIK_CE_SUM) {
      if (act == EY_ACT_TANH) MF_SHAPE_LAUNCH(MfShape<3 MF_COMMA EY_ACT_TANH MF_COMMA EY_LIK_CE_SUM>);
      MF_SHAPE_LAUNCH(MfShape<3 MF_COMMA EY_ACT_RELU MF_COMMA EY_LIK_CE_SUM>);
    }
    if (act == EY_ACT_SIGMOID) MF_SHAPE_LAUNCH(MfShape<1 MF_COMMA EY_ACT_SIGMOID MF_COMMA EY_LIK_BCE_SUM>);
    if (act == EY_ACT_TANH) MF_SHAPE_LAUNCH(MfShape<1 MF_COMMA EY_ACT_TANH MF_COMMA EY_LIK_BCE_SUM>);
    MF_SHAPE_LAUNCH(MfShape<1 MF_COMMA EY_ACT_RELU MF_COMMA EY_LIK_BCE_SUM>);
#undef MF_SHAPE_LAUNCH
  }
  if constexpr (MODE == MODE_HMC) {
    const int variant = t_ey_variant & 15;
    if (variant & 1) return mf_launch_v<MODE, 4, 0>(a, pl->n_cu, s);
    if (bf3) {
      if (a.prior_uniform && a.ntiles >= 2 && a.ntiles <= MF_PIPE_TILES && (variant & 8)) return ey_mfma32p_hmc(a, pl->n_cu, s);
      if (a.prior_uniform && a.ntiles <= MF_TRD_TILES && !(variant & 4))
        return a.da_state ? mf_launch_v<MODE, 8, 0, true, true, 2>(a, pl->n_cu, s)
                          : mf_launch_v<MODE, 8, 0, true, false, 2>(a, pl->n_cu, s);
      if (a.prior_uniform)
        return a.da_state ? mf_launch_v<MODE, 8, 0, true, true, true>(a, pl->n_cu, s)
                          : mf_launch_v<MODE, 8, 0, true, false, true>(a, pl->n_cu, s);
      if (a.ntiles <= MF_TRD_TILES && !(variant & 4)) return mf_launch_v<MODE, 8, 0, false, true, 2>(a, pl->n_cu, s);
      return mf_launch_v<MODE, 8, 0, false, true, true>(a, pl->n_cu, s);
    }
    // the headline shape (few row tiles, one Normal(m, s) prior for all parameters) has its own, leaner instantiation
    if (a.ntiles <= MF_PARK_TILES && a.prior_uniform && !(variant & 4))
      return a.da_state ? mf_launch_v<MODE, 8, MF_PARK, true, true>(a, pl->n_cu, s)
                        : mf_launch_v<MODE, 8, MF_PARK, true, false>(a, pl->n_cu, s);
  }
  if (bf3 && MODE != MODE_MH && a.ntiles <= MF_TRD_TILES && !(t_ey_variant & 4))
    return mf_launch_v<MODE, 8, 0, false, true, MODE != MODE_MH ? 2 : 1>(a, pl->n_cu, s);
  if (bf3) return mf_launch_v<MODE, 8, 0, false, true, true>(a, pl->n_cu, s);
  return mf_launch_v<MODE, 8, 0>(a, pl->n_cu, s);
}

static void mf_set_run(MfArgs& a, const EyRun* run) {
  a.n_iters = 1;
  if (run) {
    a.n_iters = run->n_iters;
    a.rec_samples = (float*)run->samples;
    a.rec_targets = (float*)run->targets;
    a.rec_accepted = (unsigned char*)run->accepted;
    a.accept_count = run->accept_count;
  }
}

int ey_mfma32_hmc(ey_plan* pl, void* theta, void* target, void* grad, const void* p0, const void* u, double step,
                  const void* step_vec, int L, const void* temp, int64_t C, uint64_t seed, uint64_t iter,
                  uint64_t chain_offset, uint32_t flags, void* accepted, void* rate, void* hcur, void* hprop,
                  hipStream_t s, const EyRun* run, const EyDA* da) {
  MfArgs a = {};
  if (da && da->state) {
    a.da_state = da->state; a.da_tab = da->table; a.da_step = (float*)da->step; a.da_n = da->n;
    a.da_final_it = da->final_it; a.da_has_eub = da->has_eub; a.da_d = da->d; a.da_logeub = da->logeub;
  }
  // while a dual averaging is attached its step vector is THE step, also once its table is used up (include/eeyore_amd.h)
  if (da && da->step) step_vec = da->step;  // the kernel reads each iteration's step where the previous one's update left it
  a.C = C; a.theta = (float*)theta; a.target = (float*)target; a.grad = (float*)grad;
  a.p0 = (const float*)p0; a.u = (const float*)u; a.step = (float)step; a.step_vec = (const float*)step_vec;
  a.L = L; a.temp = (const float*)temp; a.seed = seed; a.iter = iter; a.chain_offset = chain_offset;
  a.recompute = (flags & EY_RECOMPUTE_INITIAL_GRAD) ? 1 : 0;
  a.accepted = (unsigned char*)accepted; a.rate = (float*)rate; a.hcur = (float*)hcur; a.hprop = (float*)hprop;
  mf_set_run(a, run);
  return mf_finish(pl, a, mf_launch<MODE_HMC>(pl, a, s), s);
}

int ey_mfma32_mala(ey_plan* pl, void* theta, void* target, void* grad, const void* z, const void* u, double step,
                   const void* step_vec, const void* temp, int64_t C, uint64_t seed, uint64_t iter,
                   uint64_t chain_offset, void* accepted, void* log_rate, hipStream_t s, const EyRun* run) {
  MfArgs a = {};
  a.C = C; a.theta = (float*)theta; a.target = (float*)target; a.grad = (float*)grad;
  a.p0 = (const float*)z; a.u = (const float*)u; a.step = (float)step; a.sqrt_step = (float)sqrt(step);
  a.step_vec = (const float*)step_vec; a.temp = (const float*)temp; a.seed = seed; a.iter = iter;
  a.chain_offset = chain_offset; a.accepted = (unsigned char*)accepted; a.rate = (float*)log_rate;
  mf_set_run(a, run);
  return mf_finish(pl, a, mf_launch<MODE_MALA>(pl, a, s), s);
}

int ey_mfma32_mh(ey_plan* pl, void* theta, void* target, const void* z, const void* u, const void* scale,
                 const void* temp, int64_t C, uint64_t seed, uint64_t iter, uint64_t chain_offset, void* accepted,
                 void* log_rate, hipStream_t s, const EyRun* run) {
  MfArgs a = {};
  a.C = C; a.theta = (float*)theta; a.target = (float*)target; a.grad = (float*)theta;  // grad unused by MH
  a.p0 = (const float*)z; a.u = (const float*)u; a.scale = (const float*)scale; a.temp = (const float*)temp;
  a.seed = seed; a.iter = iter; a.chain_offset = chain_offset; a.accepted = (unsigned char*)accepted;
  a.rate = (float*)log_rate;
  mf_set_run(a, run);
  return mf_finish(pl, a, mf_launch<MODE_MH>(pl, a, s), s);
}

int ey_mfma32_log_target_grad(ey_plan* pl, const void* theta, const void* temp, int64_t C, void* target, void* grad,
                              hipStream_t s) {
  MfArgs a = {};
  a.C = C; a.theta = (float*)theta; a.target = (float*)target; a.grad = (float*)grad; a.temp = (const float*)temp;
  return mf_launch<MODE_GRAD>(pl, a, s);
}

int ey_mfma32_leapfrog(ey_plan* pl, void* theta, void* p, double step, const void* step_vec, int L, const void* temp,
                       int64_t C, void* target, void* grad, hipStream_t s) {
  MfArgs a = {};
  a.C = C; a.theta = (float*)theta; a.pio = (float*)p; a.target = (float*)target; a.grad = (float*)grad;
  a.step = (float)step; a.step_vec = (const float*)step_vec; a.L = L; a.temp = (const float*)temp;
  return mf_launch<MODE_LEAPFROG>(pl, a, s);
}

#if EY_PHASE_TIMING
extern "C" int ey_debug_ints(int* out64) {
  return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_ey_dbg), 64 * sizeof(int)) != hipSuccess;
}
extern "C" int ey_debug_wave_times(unsigned long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ey_wave_t), (size_t)n * 3 * sizeof(unsigned long long)) != hipSuccess;
}
extern "C" int ey_debug_phase_read(unsigned long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_ey_phase), 32 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[32] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_ey_phase), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
#endif
#endif  // EY_MF_PART
