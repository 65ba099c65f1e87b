// Fused value + gradient of a MID-SIZE MLP (hidden widths 33 .. 128, one or two hidden layers, a narrow output layer), f32:
// ONE launch per evaluation, one persistent 8-wave workgroup per chain at a time, the chain's weights resident in LDS for
// all of its row tiles and the weight-gradient accumulators resident in registers.
//
// The layerwise path (ey_large.hip) serves such a model as one batched GEMM launch per layer and direction; at K = 100 a
// workgroup of those products is seven k-chunks between a prologue and an epilogue of one memory round trip each, and
// MLP(20-100-100-5) ran at a quarter of the f32 matrix peak, MLP(10-100-10) at a tenth.  Here a chain's theta (<= ~60 KB) is
// staged in LDS once per evaluation and the chain's rows go through it 32 at a time:
//   * the workgroup is 2 row groups x 4 feature blocks: wave (rt, fb) owns features 32 fb .. 32 fb + 31 of every hidden
//     layer for the row tiles 2 k + rt; both groups share the weight images, each has its own activation buffers;
//   * every layer is computed transposed, H_l^T = W_l H_{l-1}^T, on v_mfma_f32_32x32x2_f32 (accumulator register 4q + j of
//     lane (c, h) <-> feature 8q + 4h + j of row c), operands read from LDS with ds_read_b128 along k (k-step (u, j) gives
//     k-slot h the index 8u + 4h + j in both operands);
//   * the narrow output layer (d_K <= 16) is not a matrix product: every wave dots its 32 features of H with the output
//     weights on the vector ALUs, the four partial logits per (row, output) meet in LDS, one wave finishes the loss;
//   * backward: dW_l = delta_{l+1}^T H_l contracts over the tile's 32 rows (16 MFMAs per 32 x 32 block pair, the pairs of a
//     layer dealt round-robin to the four waves of a group, accumulated in registers across ALL row tiles of the chain);
//     delta_l = (delta_{l+1} W_l) * act'(H_l) takes act'(H_l) from the registers the forward pass left it in and
//     overwrites H_l's LDS buffer;
//   * at the end the two row groups' accumulators meet through LDS, the prior gradient and the temperature are applied
//     (bayesian_model.py:46-50, :33-34) and the gradient is written once.
// Reference semantics: MLP.forward eeyore/models/mlp.py:45-50, CE / BCE sums eeyore/constants/constants.py:15-18 and
// eeyore/stats/loss.py:1-11, the gradient eeyore/models/log_target_model.py:15-23 (autograd there).
#include <algorithm>
#include <cstdlib>

#include "ey_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MID_NL 4  // layers at most (k_mid: two hidden layers and the output layer; k_mid32: three)

struct MidArgs {
  const float* theta;   // [C, P]
  const float* x;       // [N, d0]
  const float* y;       // [N, dK]
  const int* labels;    // [N]
  const float* mu;      // [P]
  const float* iv;      // [P]
  const float* temp;    // [C] or null
  float* lik_o;         // [C]: the untempered log-likelihood
  float* grad;          // [C, P]: gradient of the tempered log-target
  const int* tab;       // k_mid32: [P] where each parameter goes in the LDS images (k_mid32_table)
  int C, N, P, nl, lik, prior_uniform;
  float mu0, iv0;
  int dims[MID_NL + 1], woff[MID_NL], boff[MID_NL], act[MID_NL];
  // LDS carve, in floats
  int w_at[MID_NL], ldw[MID_NL];  // weights of layer l: [d_{l+1}][ldw], columns beyond d_l zero
  int b_at[MID_NL];               // bias of layer l: [roundup32(d_{l+1})], zero beyond d_{l+1} / when the layer has none
  int grp_at, grp_floats;         // the two row groups' regions
  int h_at[MID_NL], ldh[MID_NL];  // inside a region: the data tile (l = 0) and H_l / delta_l (l >= 1): [32][ldh]
  int d3_at;                      // delta of the output layer [32][16 + 4]
  int pl_at, dkp;                 // partial logits [4 waves][dkp][32 rows]
  int red_per;                    // accumulator blocks per pass of the end-of-chain reduction
  int xs_at;                      // k_mid32: the workgroup-shared image of the whole batch [32 ntiles][ldh[0]], or -1
  int total_floats;
};

__device__ __forceinline__ float mid_act(int code, float g) {
  switch (code) {
    case EY_ACT_SIGMOID: return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * g));
    case EY_ACT_TANH: return 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * g)) - 1.0f;
    case EY_ACT_RELU: return g > 0.0f ? g : 0.0f;
    default: return g;
  }
}
__device__ __forceinline__ float mid_dact(int code, float h) {
  switch (code) {
    case EY_ACT_SIGMOID: return h * (1.0f - h);
    case EY_ACT_TANH: return 1.0f - h * h;
    case EY_ACT_RELU: return h > 0.0f ? 1.0f : 0.0f;
    default: return 1.0f;
  }
}
// a whole accumulator tile through the activation / times act'(h), the (wave-uniform) switch outside the element loop
__device__ __forceinline__ void mid_act_tile(int code, f32x16& a) {
  if (code == EY_ACT_SIGMOID) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * a[r]));
  } else if (code == EY_ACT_TANH) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * a[r])) - 1.0f;
  } else if (code == EY_ACT_RELU) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = fmaxf(a[r], 0.0f);
  }
}
__device__ __forceinline__ void mid_dact_tile(int code, f32x16& d, const f32x16& hh) {
  if (code == EY_ACT_SIGMOID) {
#pragma unroll
    for (int r = 0; r < 16; ++r) d[r] *= hh[r] * (1.0f - hh[r]);
  } else if (code == EY_ACT_TANH) {
#pragma unroll
    for (int r = 0; r < 16; ++r) d[r] *= 1.0f - hh[r] * hh[r];
  } else if (code == EY_ACT_RELU) {
#pragma unroll
    for (int r = 0; r < 16; ++r) d[r] = hh[r] > 0.0f ? d[r] : 0.0f;
  }
}
__device__ __forceinline__ f32x16 mid_mfma(float a, float b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// H^T block = W H_in^T + b for output features f0 .. f0 + 31 (rows beyond d_out read the last row: their results are
// dropped where they are stored), contraction over kpad = roundup8(d_in) columns (zero beyond d_in in both images)
__device__ __forceinline__ f32x16 mid_fwd_block(const float* W, int ldw, int d_out, int f0, const float* bias, const float* Hin,
                                                int ldh, int kpad, int c, int h) {
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + f0 + 8 * q + 4 * h);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[4 * q + j] = bv[j];
  }
  const int fr = f0 + c < d_out ? f0 + c : d_out - 1;
  const float* wr = W + fr * ldw + 4 * h;
  const float* hr = Hin + c * ldh + 4 * h;
  // (the next chunk's operands are in flight while this chunk's four products issue)
  f32x4 a = *reinterpret_cast<const f32x4*>(wr), b = *reinterpret_cast<const f32x4*>(hr);
  for (int k0 = 8; k0 < kpad; k0 += 8) {
    const f32x4 an = *reinterpret_cast<const f32x4*>(wr + k0), bn = *reinterpret_cast<const f32x4*>(hr + k0);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = mid_mfma(a[j], b[j], acc);
    a = an;
    b = bn;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) acc = mid_mfma(a[j], b[j], acc);
  return acc;
}
// (delta W)^T block for input features i0 .. i0 + 31: out[i][row] = sum_f W[f][i] delta[row][f]; delta's columns beyond
// d_out are zero, so the rows of W read for them (clamped) contribute nothing
__device__ __forceinline__ f32x16 mid_dh_block(const float* W, int ldw, int d_out, int d_in, int i0, const float* D, int ldd,
                                               int c, int h) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  const int ic = i0 + c < d_in ? i0 + c : d_in - 1;
  const float* dr = D + c * ldd + 4 * h;
  const int fpad = (d_out + 7) & ~7;
  auto fetch = [&](int f0, f32x4& b, float (&a)[4]) {
    b = *reinterpret_cast<const f32x4*>(dr + f0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int f = f0 + 4 * h + j;
      a[j] = W[(f < d_out ? f : d_out - 1) * ldw + ic];
    }
  };
  f32x4 b;
  float a[4];
  fetch(0, b, a);
  for (int f0 = 8; f0 < fpad; f0 += 8) {  // (the next chunk's operands in flight under this chunk's products)
    f32x4 bn;
    float an[4];
    fetch(f0, bn, an);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = mid_mfma(a[j], b[j], acc);
    b = bn;
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = an[j];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) acc = mid_mfma(a[j], b[j], acc);
  return acc;
}
// acc[f][i] += sum over the tile's 32 rows of delta[row][f0 + .] Hprev[row][i0 + .]; returns this lane's share (its half's 16
// rows) of the column sum of delta for feature f0 + c (the bias gradient)
__device__ __forceinline__ float mid_dw_block(f32x16& acc, const float* D, int ldd, int f0, int fhi, const float* Hp, int ldh,
                                              int i0, int ihi, int c, int h) {
  const float* dr = D + (f0 + c < fhi ? f0 + c : fhi - 1);  // (columns beyond the images are read from their last one: unused outputs)
  const float* hp = Hp + (i0 + c < ihi ? i0 + c : ihi - 1);
  // (all sixteen operand pairs are in flight before the first product: one LDS round trip per block, not one per MFMA)
  dr += 4 * h * ldd;
  hp += 4 * h * ldh;
  float av[16], bv[16];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      av[4 * u + j] = dr[(8 * u + j) * ldd];
      bv[4 * u + j] = hp[(8 * u + j) * ldh];
    }
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    acc = mid_mfma(av[k], bv[k], acc);
    s += av[k];
  }
  return s;
}

// this wave's 32 x 32 block of an activation buffer, in the accumulator layout (register 4q + j <-> feature f0 + 8q + 4h + j of row c)
__device__ __forceinline__ f32x16 mid_read_block(const float* Hb, int ldh, int f0, int c, int h) {
  f32x16 v;
  const float* p = Hb + c * ldh + f0 + 4 * h;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p + 8 * q);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[4 * q + j] = t[j];
  }
  return v;
}

// BIGX: the first layer has more than 32 inputs (up to four block pairs per wave instead of one)
// MID_TIMING (diagnostic builds, tools/mid_phase.py): s_memtime sums per phase, wave 0 of every 64th workgroup
#ifndef MID_TIMING
#define MID_TIMING 0
#endif
#if MID_TIMING
__device__ unsigned long long g_mid_phase[32];
#define MT(i) do { if (mt_on) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); mt_acc[i] += n_ - mt_t; mt_t = n_; } } while (0)
extern "C" int ey_debug_mid_phase_read(unsigned long long* out32, int reset) {
  if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_mid_phase), 32 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[32] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_mid_phase), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
#else
#define MT(i) do { } while (0)
#endif
typedef const __attribute__((address_space(4))) MidArgs KA;
#define MID_ARGS() ({ KA* p_ = (KA*)__builtin_amdgcn_kernarg_segment_ptr(); asm volatile("" : "+s"(p_)); p_; })

template <bool BIGX>
__global__ void __launch_bounds__(512, 2) k_mid(MidArgs A_) {
  // The arguments are read where they are used, from the kernarg segment (scalar loads), through a pointer the compiler
  // cannot see through (MID_ARGS): hoisted to the top of the kernel its sixty integers stayed live in scalar registers for
  // the whole launch -- 530 of them spilled into vector lanes, which then spilled 640 vector registers to scratch.
  KA* A = MID_ARGS();
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rt = wave >> 2, fb = wave & 3;
  const int c = lane & 31, h = lane >> 5;
  const int nl = A->nl, dK = A->dims[nl];
  const int gtid = tid & 255;  // thread within its row group
  float* grp = smem + A->grp_at + rt * A->grp_floats;
  constexpr int S0 = BIGX ? 4 : 1;  // block-pair slots of the first layer per wave
  const int ntiles = (A->N + 31) / 32, rounds = (ntiles + 1) / 2;

#if MID_TIMING
  unsigned long long mt_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const bool mt_on = (blockIdx.x & 63) == 0 && wave == 0;
  unsigned long long mt_t = mt_on ? __builtin_amdgcn_s_memtime() : 0ull;
#endif
  for (int chain = blockIdx.x; chain < A->C; chain += gridDim.x) {
    A = MID_ARGS();
    MT(15);
    const float* th = A->theta + (size_t)chain * A->P;
    __syncthreads();  // the previous chain's reads of the images are done
    // ---- stage the chain's weights and biases
    for (int l = 0; l < nl; ++l) {
      const int din = A->dims[l], dout = A->dims[l + 1], ldw = A->ldw[l];
      float* W = smem + A->w_at[l];
      // (a row per 32 lanes: no index division; the loads of eight rows are in flight before the first store -- taken one
      // by one every element waited out its own trip to memory: 26 000 cycles per chain for 12 705 weights)
      const float* src = th + A->woff[l];
      const int kk = tid & 31;
      for (int f0 = tid >> 5; f0 < dout; f0 += 16 * 8) {
        for (int k = kk; k < ldw; k += 32) {
          float v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int f = f0 + 16 * u;
            v[u] = (k < din && f < dout) ? src[(f < dout ? f : 0) * din + (k < din ? k : 0)] : 0.0f;
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int f = f0 + 16 * u;
            if (f < dout) W[f * ldw + k] = v[u];
          }
        }
      }
      float* B = smem + A->b_at[l];
      const int bp = (dout + 31) & ~31;
      for (int e = tid; e < bp; e += 512) B[e] = (e < dout && A->boff[l] >= 0) ? th[A->boff[l] + e] : 0.0f;
    }
    f32x16 acc0[S0], acc1[4], accL;
    float db0[S0], db1[4], dbL = 0.0f, lik = 0.0f;
#pragma unroll
    for (int s = 0; s < S0; ++s) {
      db0[s] = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc0[s][r] = 0.0f;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      db1[s] = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[s][r] = 0.0f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) accL[r] = 0.0f;

    MT(0);  // staging
    for (int rd = 0; rd < rounds; ++rd) {
      A = MID_ARGS();
      const int row0 = 32 * (2 * rd + rt);
      // ---- the data tile (rows beyond N and columns beyond d0 are zeros)
      __syncthreads();
      {
        const int d0 = A->dims[0], ldx = A->ldh[0];
        float* X = grp + A->h_at[0];
        const int r = gtid >> 3;  // eight threads per row
        const bool rv = row0 + r < A->N;
        const float* xr = A->x + (size_t)(rv ? row0 + r : 0) * d0;
        for (int k = gtid & 7; k < ldx; k += 8) X[r * ldx + k] = (k < d0 && rv) ? xr[k] : 0.0f;
      }
      // this lane's row's label (CE) is fetched here, a whole round trip to memory ahead of the loss that uses it
      int lab_pre = 0;
      if (A->lik != EY_LIK_BCE_SUM && row0 + c < A->N) lab_pre = A->labels[row0 + c];
      __syncthreads();
      MT(1);  // data tile
      // ---- forward through the hidden layers (mlp.py:45-50)
#pragma unroll
      for (int l = 0; l < 2; ++l) {
        if (l < nl - 1) {
          const int din = A->dims[l], dout = A->dims[l + 1];
          if (32 * fb < ((dout + 31) & ~31)) {  // (a wave whose feature block lies beyond the layer's width has nothing here)
          f32x16 acc = mid_fwd_block(smem + A->w_at[l], A->ldw[l], dout, 32 * fb, smem + A->b_at[l], grp + A->h_at[l], A->ldh[l],
                                     (din + 7) & ~7, c, h);
          mid_act_tile(A->act[l], acc);
          float* Ho = grp + A->h_at[l + 1] + c * A->ldh[l + 1] + 32 * fb + 4 * h;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int f = 32 * fb + 8 * q + 4 * h + j;
              v[j] = f < dout ? acc[4 * q + j] : 0.0f;
            }
            *reinterpret_cast<f32x4*>(Ho + 8 * q) = v;
          }
          }
          __syncthreads();
        }
      }
      MT(2);  // forward
      A = MID_ARGS();
      // ---- the output layer: partial logits over this wave's 32 features, on the vector ALUs
      const int lt = nl - 1;  // the output layer
      {
        const bool mine = 32 * fb < ((A->dims[lt] + 31) & ~31);
        f32x16 Hl;
        if (mine) Hl = mid_read_block(grp + A->h_at[lt], A->ldh[lt], 32 * fb, c, h);
        const float* W = smem + A->w_at[lt];
        const int ldw = A->ldw[lt];
        float* PL = grp + A->pl_at + fb * A->dkp * 32;
        float sv[16];
#pragma unroll
        for (int o = 0; o < 16; ++o) {
          sv[o] = 0.0f;
          if (o < dK && mine) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f32x4 wv = *reinterpret_cast<const f32x4*>(W + o * ldw + 32 * fb + 8 * q + 4 * h);
#pragma unroll
              for (int j = 0; j < 4; ++j) sv[o] += wv[j] * Hl[4 * q + j];
            }
          }
        }
#pragma unroll
        for (int o = 0; o < 16; ++o)
          if (o < dK) sv[o] += __shfl_xor(sv[o], 32, 64);  // (independent exchanges: in flight together)
#pragma unroll
        for (int o = 0; o < 16; ++o)
          if (o < dK) PL[o * 32 + c] = sv[o];  // (both halves hold the sum and store it)
      }
      __syncthreads();
      MT(3);  // partial logits
      A = MID_ARGS();
      // ---- loss and output delta (constants.py:15-18, loss.py:1-11), by the first wave of the group: the outputs pass
      // through the row's slots of delta's image (no register arrays: d_K is a run-time number)
      if (fb == 0) {
        const float* PL = grp + A->pl_at;
        const float* bL = smem + A->b_at[lt];
        float* D3 = grp + A->d3_at + c * 20;
        const int n = row0 + c;
        const bool valid = n < A->N;
        const int code = A->act[lt];
        float out[16], mx = -3.0e38f;
#pragma unroll
        for (int o = 0; o < 16; ++o) {
          out[o] = 0.0f;
          if (o < dK)
            out[o] = (((PL[(0 * A->dkp + o) * 32 + c] + PL[(1 * A->dkp + o) * 32 + c]) + PL[(2 * A->dkp + o) * 32 + c]) +
                      PL[(3 * A->dkp + o) * 32 + c]) + bL[o];
        }
#pragma unroll
        for (int o = 0; o < 16; ++o)
          if (o < dK) {
            out[o] = mid_act(code, out[o]);
            mx = fmaxf(mx, out[o]);
          }
        float row = 0.0f;
        if (A->lik == EY_LIK_BCE_SUM) {
#pragma unroll
          for (int o = 0; o < 16; ++o)
            if (o < dK) {
              const float p = out[o], yy = valid ? A->y[(size_t)n * dK + o] : 0.0f;
              row += __logf(p) * yy + __logf(1.0f - p) * (1.0f - yy);  // naive logs (eeyore/stats/loss.py:2)
              out[o] = valid ? (yy / p - (1.0f - yy) / (1.0f - p)) * mid_dact(code, p) : 0.0f;
            }
        } else {
          const int lab = lab_pre;
          float ssum = 0.0f, olab = 0.0f, e[16];
#pragma unroll
          for (int o = 0; o < 16; ++o) {
            e[o] = 0.0f;
            if (o < dK) {
              e[o] = __expf(out[o] - mx);
              ssum += e[o];
              olab = o == lab ? out[o] : olab;
            }
          }
          row = olab - (mx + __logf(ssum));
          const float rs = 1.0f / ssum;
#pragma unroll
          for (int o = 0; o < 16; ++o)
            if (o < dK) out[o] = valid ? ((o == lab ? 1.0f : 0.0f) - e[o] * rs) * mid_dact(code, out[o]) : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<f32x4*>(D3 + 4 * q) = f32x4{out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]};
        lik += (valid && h == 0) ? row : 0.0f;
      }
      __syncthreads();
      MT(4);  // loss
      A = MID_ARGS();
      // ---- backward through the output layer: dW_K-1 (block pair (0, fb)), then delta of the last hidden layer
      f32x16 dn;
      {
        const int din = A->dims[lt], nb = (din + 31) >> 5;
        const float* D3 = grp + A->d3_at;
        const float* Hl = grp + A->h_at[lt];
        if (fb < nb) {
          const float s = mid_dw_block(accL, D3, 20, 0, 16, Hl, A->ldh[lt], 32 * fb, A->ldh[lt], c, h);
          dbL += fb == 0 ? s : 0.0f;
        }
        // delta^T = (W^T delta3^T) * act'(H): contraction over the outputs (zero beyond dK in delta3's image)
        if (fb < nb) {
          dn = mid_dh_block(smem + A->w_at[lt], A->ldw[lt], dK, din, 32 * fb, D3, 20, c, h);
          mid_dact_tile(A->act[lt - 1], dn, mid_read_block(grp + A->h_at[lt], A->ldh[lt], 32 * fb, c, h));
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int f = 32 * fb + 8 * (r >> 2) + 4 * h + (r & 3);
            dn[r] = f < din ? dn[r] : 0.0f;
          }
        }
      }
      __syncthreads();  // every wave has read H_{K-1} for its dW block: the buffer takes delta_{K-1}
      if (32 * fb < ((A->dims[lt] + 31) & ~31)) {
        float* Ho = grp + A->h_at[lt] + c * A->ldh[lt] + 32 * fb + 4 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<f32x4*>(Ho + 8 * q) = f32x4{dn[4 * q], dn[4 * q + 1], dn[4 * q + 2], dn[4 * q + 3]};
      }
      __syncthreads();
      MT(5);  // output layer backward
      A = MID_ARGS();
      // ---- backward through the second hidden layer's weights W_1 (three-layer models)
      if (nl == 3) {
        const int din = A->dims[1], dout = A->dims[2], mb = (dout + 31) >> 5, nb = (din + 31) >> 5;
        const float* D = grp + A->h_at[2];
        const float* Hp = grp + A->h_at[1];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int p = fb + 4 * s;
          if (p < mb * nb) {
            const int mm = p / nb, nn = p - mm * nb;
            const float sm = mid_dw_block(acc1[s], D, A->ldh[2], 32 * mm, A->ldh[2], Hp, A->ldh[1], 32 * nn, A->ldh[1], c, h);
            db1[s] += nn == 0 ? sm : 0.0f;
          }
        }
        if (fb < nb) {
          dn = mid_dh_block(smem + A->w_at[1], A->ldw[1], dout, din, 32 * fb, D, A->ldh[2], c, h);
          mid_dact_tile(A->act[0], dn, mid_read_block(grp + A->h_at[1], A->ldh[1], 32 * fb, c, h));
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int f = 32 * fb + 8 * (r >> 2) + 4 * h + (r & 3);
            dn[r] = f < din ? dn[r] : 0.0f;
          }
        }
        __syncthreads();
        if (fb < nb) {
          float* Ho = grp + A->h_at[1] + c * A->ldh[1] + 32 * fb + 4 * h;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<f32x4*>(Ho + 8 * q) = f32x4{dn[4 * q], dn[4 * q + 1], dn[4 * q + 2], dn[4 * q + 3]};
        }
        __syncthreads();
      }
      MT(6);  // second hidden layer backward
      A = MID_ARGS();
      // ---- the first layer's weights: dW_0 = delta_1^T x
      {
        const int din = A->dims[0], dout = A->dims[1], mb = (dout + 31) >> 5, nb = (din + 31) >> 5;
        const float* D = grp + A->h_at[1];
        const float* Xp = grp + A->h_at[0];
#pragma unroll
        for (int s = 0; s < S0; ++s) {
          const int p = fb + 4 * s;
          if (p < mb * nb) {
            const int mm = p / nb, nn = p - mm * nb;
            const float sm = mid_dw_block(acc0[s], D, A->ldh[1], 32 * mm, A->ldh[1], Xp, A->ldh[0], 32 * nn, A->ldh[0], c, h);
            db0[s] += nn == 0 ? sm : 0.0f;
          }
        }
      }
      MT(7);  // first layer's weights
    }  // row tiles

    A = MID_ARGS();
    // ---- the two row groups' sums meet (group 1 -> LDS -> group 0), prior gradient and temperature, write-out
    const float tsc = A->temp ? A->temp[chain] : 1.0f;
    float* gout = A->grad + (size_t)chain * A->P;
    float* red = smem + A->grp_at;  // both regions are free now: [4 waves][17][64 lanes]
    // (as many accumulator blocks per pass as the two regions hold, A->red_per: two barriers per pass, not per block)
    constexpr int NS = S0 + 5;  // slots: the first layer's, the second layer's four, the output layer's
    auto slot_on = [&](int i) { return i < S0 || i == NS - 1 || nl == 3; };
    auto red_put = [&](int pos, const f32x16& acc, float dbv) {
      float* r = red + pos * (4 * 17 * 64) + fb * 17 * 64 + lane;
#pragma unroll
      for (int k = 0; k < 16; ++k) r[k * 64] = acc[k];
      r[16 * 64] = dbv;
    };
    auto red_get = [&](int pos, f32x16& acc, float& dbv) {
      const float* r = red + pos * (4 * 17 * 64) + fb * 17 * 64 + lane;
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[k] += r[k * 64];
      dbv += r[16 * 64];
    };
    const int per = A->red_per;
    for (int lo = 0; lo < NS; lo += per) {
      __syncthreads();
      if (rt == 1) {
#pragma unroll
        for (int i = 0; i < NS; ++i)
          if (i >= lo && i < lo + per && slot_on(i)) {
            if (i < S0) red_put(i - lo, acc0[i < S0 ? i : 0], db0[i < S0 ? i : 0]);
            else if (i < S0 + 4) red_put(i - lo, acc1[(i - S0) & 3], db1[(i - S0) & 3]);
            else red_put(i - lo, accL, dbL);
          }
      }
      __syncthreads();
      if (rt == 0) {
#pragma unroll
        for (int i = 0; i < NS; ++i)
          if (i >= lo && i < lo + per && slot_on(i)) {
            if (i < S0) red_get(i - lo, acc0[i < S0 ? i : 0], db0[i < S0 ? i : 0]);
            else if (i < S0 + 4) red_get(i - lo, acc1[(i - S0) & 3], db1[(i - S0) & 3]);
            else red_get(i - lo, accL, dbL);
          }
      }
    }
    A = MID_ARGS();
    auto emit = [&](int l, int mm, int nn, const f32x16& acc, float dbv) {
      const int din = A->dims[l], dout = A->dims[l + 1];
      const float* W = smem + A->w_at[l];
      const int i = 32 * nn + c;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int f = 32 * mm + 8 * (r >> 2) + 4 * h + (r & 3);
        if (f < dout && i < din) {
          const int idx = A->woff[l] + f * din + i;
          const float m_ = A->prior_uniform ? A->mu0 : A->mu[idx], i_ = A->prior_uniform ? A->iv0 : A->iv[idx];
          gout[idx] = (acc[r] - (W[f * A->ldw[l] + i] - m_) * i_) * tsc;
        }
      }
      if (nn == 0 && A->boff[l] >= 0) {
        const float tot = dbv + __shfl_xor(dbv, 32, 64);
        const int f = 32 * mm + c;
        if (h == 0 && f < dout) {
          const int idx = A->boff[l] + f;
          const float m_ = A->prior_uniform ? A->mu0 : A->mu[idx], i_ = A->prior_uniform ? A->iv0 : A->iv[idx];
          gout[idx] = (tot - (smem[A->b_at[l] + f] - m_) * i_) * tsc;
        }
      }
    };
    {
      const int mb = (A->dims[1] + 31) >> 5, nb = (A->dims[0] + 31) >> 5;
#pragma unroll
      for (int s = 0; s < S0; ++s) {
        const int p = fb + 4 * s;
        if (rt == 0 && p < mb * nb) emit(0, p / nb, p % nb, acc0[s], db0[s]);
      }
    }
    if (nl == 3) {
      const int mb = (A->dims[2] + 31) >> 5, nb = (A->dims[1] + 31) >> 5;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int p = fb + 4 * s;
        if (rt == 0 && p < mb * nb) emit(1, p / nb, p % nb, acc1[s], db1[s]);
      }
    }
    {
      const int nb = (A->dims[nl - 1] + 31) >> 5;
      if (rt == 0 && fb < nb) {
        // the output layer's block: rows of the accumulator are outputs (f < dK), its bias sums sit in lanes c < dK
        const int l = nl - 1, din = A->dims[l];
        const float* W = smem + A->w_at[l];
        const int i = 32 * fb + c;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int f = 8 * (r >> 2) + 4 * h + (r & 3);
          if (f < dK && i < din) {
            const int idx = A->woff[l] + f * din + i;
            const float m_ = A->prior_uniform ? A->mu0 : A->mu[idx], i_ = A->prior_uniform ? A->iv0 : A->iv[idx];
            gout[idx] = (accL[r] - (W[f * A->ldw[l] + i] - m_) * i_) * tsc;
          }
        }
        if (fb == 0 && A->boff[l] >= 0) {
          const float tot = dbL + __shfl_xor(dbL, 32, 64);
          if (h == 0 && c < dK) {
            const int idx = A->boff[l] + c;
            const float m_ = A->prior_uniform ? A->mu0 : A->mu[idx], i_ = A->prior_uniform ? A->iv0 : A->iv[idx];
            gout[idx] = (tot - (smem[A->b_at[l] + c] - m_) * i_) * tsc;
          }
        }
      }
    }
    // ---- the log-likelihood: lanes of the two groups' first waves, in a fixed order
    __syncthreads();
    if (fb == 0) {
      float v = lik;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      if (lane == 0) red[rt] = v;
    }
    __syncthreads();
    if (tid == 0) A->lik_o[chain] = red[0] + red[1];
    MT(8);  // combine + write-out
#if MID_TIMING
    if (mt_on && lane == 0) atomicAdd(&g_mid_phase[31], 1ull);
#endif
  }
#if MID_TIMING
  if (mt_on && lane == 0)
    for (int i = 0; i < 16; ++i) atomicAdd(&g_mid_phase[i], mt_acc[i]);
#endif
}

// k_mid32's activation buffers are [32 rows][32 columns] with the 16-byte chunk of a row XORed with the row's low three bits
// (36-float rows did not fit eight waves' buffers in the CU's LDS): a 16-byte read of one chunk by 32 consecutive rows, and a
// 4-byte read of one row by 32 consecutive columns, both touch every bank once.
__device__ __forceinline__ int hsw(int row, int col) { return row * 32 + ((((col >> 2) ^ row) & 7) << 2) + (col & 3); }
// H^T block = W H_in^T + b (one feature block: f0 = 0), H_in swizzled (SWZ) or the data tile / image with row stride ldh
template <bool SWZ>
__device__ __forceinline__ f32x16 m32_fwd(const float* W, int ldw, int d_out, const float* bias, const float* Hin, int ldh,
                                          int kpad, int c, int h) {
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 8 * q + 4 * h);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[4 * q + j] = bv[j];
  }
  const float* wr = W + (c < d_out ? c : d_out - 1) * ldw + 4 * h;
  for (int k0 = 0; k0 < kpad; k0 += 8) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(wr + k0);
    const f32x4 b = *reinterpret_cast<const f32x4*>(SWZ ? Hin + hsw(c, k0 + 4 * h) : Hin + c * ldh + 4 * h + k0);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = mid_mfma(a[j], b[j], acc);
  }
  return acc;
}
// store / read a tile in the accumulator layout to / from a swizzled buffer (columns beyond `valid` as zeros)
__device__ __forceinline__ void m32_store(float* Hb, f32x16& v, int valid, int c, int h) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 t;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      t[j] = 8 * q + 4 * h + j < valid ? v[4 * q + j] : 0.0f;
      v[4 * q + j] = t[j];
    }
    *reinterpret_cast<f32x4*>(Hb + hsw(c, 8 * q + 4 * h)) = t;
  }
}
__device__ __forceinline__ f32x16 m32_read(const float* Hb, int c, int h) {
  f32x16 v;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(Hb + hsw(c, 8 * q + 4 * h));
#pragma unroll
    for (int j = 0; j < 4; ++j) v[4 * q + j] = t[j];
  }
  return v;
}
// (delta W)^T: out[i][row] = sum_f W[f][i] delta[row][f]; delta in a swizzled buffer (DSWZ) or the output delta's [32][20] image
template <bool DSWZ>
__device__ __forceinline__ f32x16 m32_dh(const float* W, int ldw, int d_out, int d_in, const float* D, int c, int h) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  const int ic = c < d_in ? c : d_in - 1;
  const int fpad = (d_out + 7) & ~7;
  for (int f0 = 0; f0 < fpad; f0 += 8) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(DSWZ ? D + hsw(c, f0 + 4 * h) : D + c * 20 + 4 * h + f0);
    float a[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int f = f0 + 4 * h + j;
      a[j] = W[(f < d_out ? f : d_out - 1) * ldw + ic];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = mid_mfma(a[j], b[j], acc);
  }
  return acc;
}
// acc[f][i] += sum_rows delta[row][f] Hprev[row][i0 + i]; delta swizzled (DSWZ) or the [32][20] image (columns clamped to
// 15), Hprev swizzled (HSWZ) or a linear buffer with row stride ldh and ihi columns; returns this lane's share of delta's column sum
template <bool DSWZ, bool HSWZ>
__device__ __forceinline__ float m32_dw(f32x16& acc, const float* D, const float* Hp, int ldh, int i0, int ihi, int c, int h) {
  const int fc = DSWZ ? c : (c < 16 ? c : 15);
  const int icol = HSWZ ? c : (i0 + c < ihi ? i0 + c : ihi - 1);
  // row 8u + 4h + j of a swizzled buffer: (4h + j) * 32 + (((col >> 2) ^ (4h + j)) & 7) * 4 + (col & 3), plus 256 u -- four
  // base addresses per operand, the rest immediate offsets
  const float* ab[4];
  const float* bb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int rj = 4 * h + j;
    ab[j] = DSWZ ? D + rj * 32 + ((((fc >> 2) ^ rj) & 7) << 2) + (fc & 3) : D + rj * 20 + fc;
    bb[j] = HSWZ ? Hp + rj * 32 + ((((icol >> 2) ^ rj) & 7) << 2) + (icol & 3) : Hp + rj * ldh + icol;
  }
  float av[16], bv[16];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      av[4 * u + j] = DSWZ ? ab[j][256 * u] : ab[j][160 * u];
      bv[4 * u + j] = HSWZ ? bb[j][256 * u] : bb[j][8 * u * ldh];
    }
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    acc = mid_mfma(av[k], bv[k], acc);
    s += av[k];
  }
  return s;
}

// ---- k_mid32: the same building blocks for NARROW, DEEPER models (every hidden width <= 32, up to three hidden layers, up
// to 64 inputs, d_K <= 16): with one feature block per layer a row tile is ONE wave's work from the data tile to its share of
// every weight gradient, so the eight waves of the chain's workgroup take row tiles 8k + w and meet only at the ends of
// the chain (weights staged in LDS before, the waves' accumulators summed through LDS behind): no barrier, no shared
// activation buffer inside a round -- the two waves of a SIMD are in different phases by themselves.  The last hidden layer's
// tile goes from its accumulator registers straight into the output layer's dots; every wave finishes its own rows' loss.
// (Three hidden layers and more than 16 inputs are the shapes fused16 does not take: VERDICT r4 item 8.)
__device__ __forceinline__ void mid_wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // LDS operations of one wave execute in order: a compiler fence
  __builtin_amdgcn_wave_barrier();
}
// where in the LDS images each parameter of theta goes (k_mid32 stages a chain with all its loads in flight at once: taken
// row by row every element waited out its own trip to memory, a dozen serial round trips per chain)
__global__ void k_mid32_table(MidArgs A, int* __restrict__ tab) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= A.P) return;
  int dst = -1;
  for (int l = 0; l < A.nl; ++l) {
    const int din = A.dims[l], dout = A.dims[l + 1];
    if (e >= A.woff[l] && e < A.woff[l] + din * dout) {
      const int f = (e - A.woff[l]) / din, k = (e - A.woff[l]) - f * din;
      dst = A.w_at[l] + f * A.ldw[l] + k;
    }
    if (A.boff[l] >= 0 && e >= A.boff[l] && e < A.boff[l] + dout) dst = A.b_at[l] + (e - A.boff[l]);
  }
  tab[e] = dst;
}
#define MID32_SPT 10  // parameters per thread at most (P <= 4720 for 64 inputs, three hidden layers of 32, 16 outputs)
template <int NB0>  // 32-wide input blocks of the first layer: 1 (d_0 <= 32) or 2
__global__ void __launch_bounds__(512, 2) k_mid32(MidArgs A_) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  KA* A = MID_ARGS();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5;
  const int nl = A->nl, dK = A->dims[nl];
  float* wr_ = smem + A->grp_at + wave * A->grp_floats;  // this wave's activation buffers
  const int ntiles = (A->N + 31) / 32, rounds = (ntiles + 7) / 8;
#if MID_TIMING
  unsigned long long mt_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const bool mt_on = (blockIdx.x & 63) == 0 && wave == 0;
  unsigned long long mt_t = mt_on ? __builtin_amdgcn_s_memtime() : 0ull;
#endif
  for (int e = tid; e < A->grp_at; e += 512) smem[e] = 0.0f;  // the images' padding (and a missing bias) stays zero for the launch
  __syncthreads();
  if (A->xs_at >= 0) {  // the whole batch, zero-padded to whole tiles and to the row stride, once per launch
    const int d0 = A->dims[0], ldx = A->ldh[0];
    float* XS = smem + A->xs_at;
    for (int r = tid >> 3; r < 32 * ntiles; r += 64) {
      const bool rv = r < A->N;
      const float* xr = A->x + (size_t)(rv ? r : 0) * d0;
      for (int k = tid & 7; k < ldx; k += 8) XS[r * ldx + k] = (k < d0 && rv) ? xr[k] : 0.0f;
    }
  }
  // a chain's parameters are fetched while the chain before it sums and writes its gradient (the end of a chain is barriers
  // and LDS round trips: the fetch's trip to memory costs nothing there and 2 500 cycles in front of the tiles)
  float vnx[MID32_SPT];
  auto fetch_theta = [&](int ch) {
    const float* thn = MID_ARGS()->theta + (size_t)ch * MID_ARGS()->P;
    const int P = MID_ARGS()->P;
#pragma unroll
    for (int t = 0; t < MID32_SPT; ++t) {
      const int e = tid + 512 * t;
      vnx[t] = e < P ? thn[e] : 0.0f;
    }
  };
  if ((int)blockIdx.x < A->C) fetch_theta(blockIdx.x);
  for (int chain = blockIdx.x; chain < A->C; chain += gridDim.x) {
    A = MID_ARGS();
    __syncthreads();
    {
      int d[MID32_SPT];
      const int P = A->P;
#pragma unroll
      for (int t = 0; t < MID32_SPT; ++t) {
        const int e = tid + 512 * t;
        d[t] = e < P ? A->tab[e] : -1;
      }
#pragma unroll
      for (int t = 0; t < MID32_SPT; ++t)
        if (d[t] >= 0) smem[d[t]] = vnx[t];
    }
    __syncthreads();
    MT(0);  // staging
#if MID_TIMING
    const unsigned long long mt_w0 = __builtin_amdgcn_s_memtime();  // every wave of a timed workgroup: its tiles' duration
#endif
    f32x16 accF[NB0], accH[2], accL;
    float dbF = 0.0f, dbH[2] = {0.0f, 0.0f}, dbL = 0.0f, lik = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int b = 0; b < NB0; ++b) accF[b][r] = 0.0f;
      accH[0][r] = 0.0f;
      accH[1][r] = 0.0f;
      accL[r] = 0.0f;
    }
    for (int rd = 0; rd < rounds; ++rd) {
      const int tile = 8 * rd + wave;
      if (tile >= ntiles) break;  // (wave-uniform; nothing below synchronises with another wave)
      A = MID_ARGS();
      const int row0 = 32 * tile;
      // ---- the data tile: rows of the workgroup's image of the whole batch (staged once per launch), or two lanes per row
      const float* X = A->xs_at >= 0 ? smem + A->xs_at + row0 * A->ldh[0] : wr_ + A->h_at[0];
      if (A->xs_at < 0) {
        const int d0 = A->dims[0], ldx = A->ldh[0];
        float* Xw = wr_ + A->h_at[0];
        const int r = lane >> 1;
        const bool rv = row0 + r < A->N;
        const float* xr = A->x + (size_t)(rv ? row0 + r : 0) * d0;
        for (int k = lane & 1; k < ldx; k += 2) Xw[r * ldx + k] = (k < d0 && rv) ? xr[k] : 0.0f;
      }
      int lab_pre = 0;
      if (A->lik != EY_LIK_BCE_SUM && row0 + c < A->N) lab_pre = A->labels[row0 + c];
      mid_wave_fence();
      // ---- forward (mlp.py:45-50); the last hidden layer's tile stays in hv
      f32x16 hv;
#pragma unroll
      for (int l = 0; l < 3; ++l) {
        if (l < nl - 1) {
          const int din = A->dims[l], dout = A->dims[l + 1];
          if (l == 0) hv = m32_fwd<false>(smem + A->w_at[0], A->ldw[0], dout, smem + A->b_at[0], X, A->ldh[0], (din + 7) & ~7, c, h);
          else hv = m32_fwd<true>(smem + A->w_at[l], A->ldw[l], dout, smem + A->b_at[l], wr_ + A->h_at[l], 0, (din + 7) & ~7, c, h);
          mid_act_tile(A->act[l], hv);
          m32_store(wr_ + A->h_at[l + 1], hv, dout, c, h);
          mid_wave_fence();
        }
      }
      MT(3);  // data tile, forward through the hidden layers
      A = MID_ARGS();
      // ---- output layer, loss and delta (constants.py:15-18, loss.py:1-11): every lane holds its row's outputs
      const int lt = nl - 1;
      float out[16];
      {
        const float* W = smem + A->w_at[lt];
        const int ldw = A->ldw[lt];
        const float* bL = smem + A->b_at[lt];
#pragma unroll
        for (int o = 0; o < 16; ++o) {
          out[o] = 0.0f;
          if (o < dK) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f32x4 wv = *reinterpret_cast<const f32x4*>(W + o * ldw + 8 * q + 4 * h);
#pragma unroll
              for (int j = 0; j < 4; ++j) out[o] += wv[j] * hv[4 * q + j];
            }
          }
        }
#pragma unroll
        for (int o = 0; o < 16; ++o)
          if (o < dK) out[o] = (out[o] + __shfl_xor(out[o], 32, 64)) + bL[o];
        const int n = row0 + c;
        const bool valid = n < A->N;
        const int code = A->act[lt];
        float mx = -3.0e38f;
        if (code != EY_ACT_NONE) {  // (wave-uniform: the usual CE head has no output activation)
#pragma unroll
          for (int o = 0; o < 16; ++o)
            if (o < dK) out[o] = mid_act(code, out[o]);
        }
#pragma unroll
        for (int o = 0; o < 16; ++o)
          if (o < dK) mx = fmaxf(mx, out[o]);
        float row = 0.0f;
        if (A->lik == EY_LIK_BCE_SUM) {
#pragma unroll
          for (int o = 0; o < 16; ++o)
            if (o < dK) {
              const float p = out[o], yy = valid ? A->y[(size_t)n * dK + o] : 0.0f;
              row += __logf(p) * yy + __logf(1.0f - p) * (1.0f - yy);  // naive logs (eeyore/stats/loss.py:2)
              out[o] = valid ? (yy / p - (1.0f - yy) / (1.0f - p)) * mid_dact(code, p) : 0.0f;
            }
        } else {
          const int lab = lab_pre;
          float ssum = 0.0f, olab = 0.0f, e[16];
#pragma unroll
          for (int o = 0; o < 16; ++o) {
            e[o] = 0.0f;
            if (o < dK) {
              e[o] = __expf(out[o] - mx);
              ssum += e[o];
              olab = o == lab ? out[o] : olab;
            }
          }
          row = olab - (mx + __logf(ssum));
          const float rs = 1.0f / ssum;
#pragma unroll
          for (int o = 0; o < 16; ++o)
            if (o < dK) out[o] = valid ? ((o == lab ? 1.0f : 0.0f) - e[o] * rs) * mid_dact(code, out[o]) : 0.0f;
        }
        lik += (valid && h == 0) ? row : 0.0f;
        float* D3 = wr_ + A->d3_at + c * 20;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<f32x4*>(D3 + 4 * q) = f32x4{out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]};
      }
      mid_wave_fence();
      MT(2);  // forward, loss
      A = MID_ARGS();
      // ---- backward: the output layer's weights, then every hidden layer from the last to the second, then the first
      {
        const int din = A->dims[lt];
        const float* D3 = wr_ + A->d3_at;
        float* Hl = wr_ + A->h_at[lt];
        dbL += m32_dw<false, true>(accL, D3, Hl, 0, 0, 32, c, h);
        f32x16 dn = m32_dh<false>(smem + A->w_at[lt], A->ldw[lt], dK, din, D3, c, h);
        mid_dact_tile(A->act[lt - 1], dn, hv);
        mid_wave_fence();
        m32_store(Hl, dn, din, c, h);
        mid_wave_fence();
      }
      MT(4);  // output layer backward
#pragma unroll
      for (int l = 2; l >= 1; --l) {
        if (l < nl - 1) {  // hidden-to-hidden weights W_l: delta_{l+1} lies in H_{l+1}'s buffer, H_l in its own
          const int din = A->dims[l], dout = A->dims[l + 1];
          const float* D = wr_ + A->h_at[l + 1];
          float* Hp = wr_ + A->h_at[l];
          dbH[l - 1] += m32_dw<true, true>(accH[l - 1], D, Hp, 0, 0, 32, c, h);
          f32x16 dn = m32_dh<true>(smem + A->w_at[l], A->ldw[l], dout, din, D, c, h);
          mid_dact_tile(A->act[l - 1], dn, m32_read(Hp, c, h));
          mid_wave_fence();
          m32_store(Hp, dn, din, c, h);
          mid_wave_fence();
        }
      }
      MT(5);  // hidden layers backward
      {
        const float* D = wr_ + A->h_at[1];
#pragma unroll
        for (int b = 0; b < NB0; ++b) {
          const float sm = m32_dw<true, false>(accF[b], D, X, A->ldh[0], 32 * b, A->ldh[0], c, h);
          dbF += b == 0 ? sm : 0.0f;
        }
      }
      mid_wave_fence();
      MT(6);  // first layer's dW
    }  // row tiles of this wave
    MT(1);  // (waves without a tile)
#if MID_TIMING
    if ((blockIdx.x & 63) == 0 && lane == 0) atomicAdd(&g_mid_phase[16 + wave], __builtin_amdgcn_s_memtime() - mt_w0);
#endif

    if (chain + (int)gridDim.x < A->C) fetch_theta(chain + gridDim.x);  // (uniform)
    // ---- the eight waves' sums meet: slot by slot through LDS, wave (slot mod 8) adds the copies in wave order and writes
    // that slot's part of the gradient (prior gradient and temperature applied, bayesian_model.py:46-50, :33-34)
    A = MID_ARGS();
    const float tsc = A->temp ? A->temp[chain] : 1.0f;
    float* gout = A->grad + (size_t)chain * A->P;
    float* red = smem + A->grp_at;  // [8 waves][17][64]
    constexpr int NS = NB0 + 3;     // slots: the first layer's blocks, W_1, W_2, the output layer
    // every wave leaves its accumulators of the pass's slots in LDS; then wave w adds the eight copies (in wave order) of
    // accumulator registers 2w and 2w + 1 of every slot -- and wave 0 the bias sums -- and writes those entries: the whole
    // workgroup sums and writes in parallel, two barriers per pass of A->red_per slots
    auto slot_layer = [&](int s_) { return s_ < NB0 ? 0 : (s_ == NB0 ? 1 : (s_ == NB0 + 1 ? 2 : nl - 1)); };
    auto slot_on = [&](int s_) { return s_ < NB0 || s_ == NS - 1 || (s_ == NB0 && nl >= 3) || (s_ == NB0 + 1 && nl >= 4); };
    auto put = [&](int pos, const f32x16& acc, float dbv) {
      float* r = red + (pos * 8 + wave) * 17 * 64 + lane;
#pragma unroll
      for (int k = 0; k < 16; ++k) r[k * 64] = acc[k];
      r[16 * 64] = dbv;
    };
    auto emit = [&](int pos, int l, int nn) {
      const int din = A->dims[l], dout = A->dims[l + 1];
      const float* W = smem + A->w_at[l];
      const int i = 32 * nn + c;
      const float* r0 = red + pos * 8 * 17 * 64 + lane;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int k = 2 * wave + kk;
        float v = 0.0f;
        for (int w = 0; w < 8; ++w) v += r0[(w * 17 + k) * 64];
        const int f = 8 * (k >> 2) + 4 * h + (k & 3);
        if (f < dout && i < din) {
          const int idx = A->woff[l] + f * din + i;
          const float m_ = A->prior_uniform ? A->mu0 : A->mu[idx], i_ = A->prior_uniform ? A->iv0 : A->iv[idx];
          gout[idx] = (v - (W[f * A->ldw[l] + i] - m_) * i_) * tsc;
        }
      }
      if (wave == (pos & 7) && nn == 0 && A->boff[l] >= 0) {  // (the slots' bias sums dealt over the waves, not all to wave 0)
        float dbv = 0.0f;
        for (int w = 0; w < 8; ++w) dbv += r0[(w * 17 + 16) * 64];
        const float tot = dbv + __shfl_xor(dbv, 32, 64);
        if (h == 0 && c < dout) {
          const int idx = A->boff[l] + c;
          const float m_ = A->prior_uniform ? A->mu0 : A->mu[idx], i_ = A->prior_uniform ? A->iv0 : A->iv[idx];
          gout[idx] = (tot - (smem[A->b_at[l] + c] - m_) * i_) * tsc;
        }
      }
    };
    const int per = A->red_per;
    MT(8);  // (temperature, pointers)
    for (int lo = 0; lo < NS; lo += per) {
      __syncthreads();
      MT(9);  // the barrier in front of a pass (the first one: waiting for the slowest wave's tiles)
#pragma unroll
      for (int s_ = 0; s_ < NS; ++s_)
        if (s_ >= lo && s_ < lo + per && slot_on(s_)) {
          if (s_ < NB0) put(s_ - lo, accF[s_ < NB0 ? s_ : 0], s_ == 0 ? dbF : 0.0f);
          else if (s_ == NB0) put(s_ - lo, accH[0], dbH[0]);
          else if (s_ == NB0 + 1) put(s_ - lo, accH[1], dbH[1]);
          else put(s_ - lo, accL, dbL);
        }
      MT(10);  // accumulators to LDS
      __syncthreads();
      MT(11);  // the barrier behind them
#pragma unroll
      for (int s_ = 0; s_ < NS; ++s_)
        if (s_ >= lo && s_ < lo + per && slot_on(s_)) emit(s_ - lo, slot_layer(s_), s_ < NB0 ? s_ : 0);
      MT(12);  // sums, prior gradient, stores
    }
    // ---- the log-likelihood, in wave order
    __syncthreads();
    {
      float v = lik;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      if (lane == 0) red[wave] = v;
    }
    __syncthreads();
    if (tid == 0) A->lik_o[chain] = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
    MT(13);  // the log-likelihood
#if MID_TIMING
    if (mt_on && lane == 0) atomicAdd(&g_mid_phase[31], 1ull);
#endif
  }
#if MID_TIMING
  if (mt_on && lane == 0)
    for (int i = 0; i < 16; ++i) atomicAdd(&g_mid_phase[i], mt_acc[i]);
#endif
}

// ------------------------------------------------------------------------------------------------------- host side
static int mid_ld(int d) {  // a row stride (floats) whose 16-byte reads by 32 consecutive rows miss each other's banks
  int s = (d + 7) & ~7;
  if (((s >> 2) & 1) == 0) s += 4;
  return s;
}
// the carve for a model, or false when this kernel does not serve it
static bool mid_plan(const EyModel& m, MidArgs& a) {
  if (m.nl < 2 || m.nl > 3) return false;
  const int nl = m.nl, dK = m.dims[nl];
  if (dK > 16 || m.dims[0] > 128) return false;
  int widest = 0;
  for (int l = 1; l < nl; ++l) {
    if (m.dims[l] > 128) return false;
    widest = std::max(widest, m.dims[l]);
  }
  if (widest <= 32) return false;  // one feature block: three of the four waves of a group would idle (other kernels serve those)
  int at = 0;
  for (int l = 0; l < nl; ++l) {
    // (the output layer's rows are read 32 features per wave by the partial logits: as wide as the hidden buffers, zero-filled)
    a.ldw[l] = l == nl - 1 ? ((m.dims[l] + 31) & ~31) + 4 : mid_ld(m.dims[l]);
    a.w_at[l] = at;
    at += m.dims[l + 1] * a.ldw[l];
    at = (at + 3) & ~3;
  }
  for (int l = 0; l < nl; ++l) {
    a.b_at[l] = at;
    at += (m.dims[l + 1] + 31) & ~31;
  }
  a.grp_at = at;
  int g = 0;
  a.ldh[0] = mid_ld(m.dims[0]);
  a.h_at[0] = g;
  g += 32 * a.ldh[0];
  for (int l = 1; l < nl; ++l) {
    a.ldh[l] = ((m.dims[l] + 31) & ~31) + 4;
    a.h_at[l] = g;
    g += 32 * a.ldh[l];
  }
  a.d3_at = g;
  g += 32 * 20;
  a.dkp = (dK + 3) & ~3;
  a.pl_at = g;
  g += 4 * a.dkp * 32;
  g = std::max(g, 4 * 17 * 64 / 2 + 64);  // the end-of-chain reduction scratch spans both regions: at least one block per pass
  g = (g + 3) & ~3;
  a.grp_floats = g;
  a.total_floats = at + 2 * g;
  a.red_per = std::max(1, (2 * g - 8) / (4 * 17 * 64));
  return (size_t)a.total_floats * sizeof(float) <= 160 * 1024;
}

// the carve of k_mid32 (one wave per row tile): per-wave buffers for the data tile, every hidden layer and the output delta
static bool mid32_plan(const EyModel& m, MidArgs& a) {
  if (m.nl < 2 || m.nl > 4) return false;
  const int nl = m.nl, dK = m.dims[nl];
  if (dK > 16 || m.dims[0] > 64) return false;
  for (int l = 1; l < nl; ++l)
    if (m.dims[l] > 32) return false;
  int at = 0;
  for (int l = 0; l < nl; ++l) {
    a.ldw[l] = l == 0 ? mid_ld(m.dims[0]) : 36;
    a.w_at[l] = at;
    at += m.dims[l + 1] * a.ldw[l];
    at = (at + 3) & ~3;
  }
  for (int l = 0; l < nl; ++l) {
    a.b_at[l] = at;
    at += 32;
  }
  a.ldh[0] = mid_ld(m.dims[0]);
  const int ntiles = (m.N + 31) / 32;
  // the data: one image of the whole batch for the workgroup when that fits beside everything else, else a tile per wave
  for (int shared = 1; shared >= 0; --shared) {
    int base = at;
    a.xs_at = -1;
    if (shared) {
      a.xs_at = base;
      base += 32 * ntiles * a.ldh[0];
    }
    a.grp_at = base;
    int g = 0;
    a.h_at[0] = g;
    if (!shared) g += 32 * a.ldh[0];
    for (int l = 1; l < nl; ++l) {
      a.ldh[l] = 32;
      a.h_at[l] = g;
      g += 32 * 32;
    }
    a.d3_at = g;
    g += 32 * 20;
    g = std::max(g, 17 * 64 + 8);  // the end-of-chain reduction scratch: [8 waves][17][64] over the eight regions
    g = (g + 3) & ~3;
    a.grp_floats = g;
    a.pl_at = 0; a.dkp = 0;
    a.red_per = std::max(1, (8 * g) / (8 * 17 * 64));  // accumulator slots per pass of the end-of-chain reduction
    a.total_floats = base + 8 * g;
    if ((size_t)a.total_floats * sizeof(float) <= 160 * 1024) return true;
  }
  return false;
}
bool ey_mid32_supports(const ey_plan* pl) {
  if (pl->dtype != EY_F32) return false;
  MidArgs a = {};
  return mid32_plan(pl->m, a);
}
static void mid_fill(ey_plan* pl, MidArgs& a, const float* theta, const float* temp, int C, float* lik_o, float* grad) {
  const EyModel& m = pl->m;
  a.theta = theta; a.x = (const float*)m.x; a.y = (const float*)m.y; a.labels = m.labels; a.mu = (const float*)m.mu;
  a.iv = (const float*)m.inv_var; a.temp = temp; a.lik_o = lik_o; a.grad = grad;
  a.C = C; a.N = m.N; a.P = m.P; a.nl = m.nl; a.lik = m.lik;
  a.prior_uniform = pl->prior_uniform ? 1 : 0; a.mu0 = (float)pl->prior_mu0; a.iv0 = (float)pl->prior_iv0;
  for (int l = 0; l <= m.nl; ++l) a.dims[l] = m.dims[l];
  for (int l = 0; l < m.nl; ++l) { a.woff[l] = m.woff[l]; a.boff[l] = m.boff[l]; a.act[l] = m.act[l]; }
}
int ey_mid32_eval(ey_plan* pl, const float* theta, const float* temp, int C, float* lik_o, float* grad, void* scratch,
                  hipStream_t s) {
  MidArgs a = {};
  if (!mid32_plan(pl->m, a)) EY_FAIL(EY_ERR_UNSUPPORTED, "ey_mid32_eval: model not served by the fused narrow-model kernel");
  mid_fill(pl, a, theta, temp, C, lik_o, grad);
  a.tab = (const int*)scratch;  // (P integers of the caller's activation workspace, which this path does not use otherwise)
  hipLaunchKernelGGL(k_mid32_table, dim3((a.P + 255) / 256), dim3(256), 0, s, a, (int*)scratch);
  const size_t bytes = (size_t)a.total_floats * sizeof(float);
  const unsigned grid = (unsigned)std::min<int64_t>(C, pl->n_cu > 0 ? pl->n_cu : 256);
  if (pl->m.dims[0] > 32) {
    EY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mid32<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(k_mid32<2>, dim3(grid), dim3(512), bytes, s, a);
  } else {
    EY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mid32<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(k_mid32<1>, dim3(grid), dim3(512), bytes, s, a);
  }
  EY_HIP(hipGetLastError());
  return EY_OK;
}

bool ey_mid_supports(const ey_plan* pl) {
  if (pl->dtype != EY_F32) return false;
  MidArgs a = {};
  return mid_plan(pl->m, a);
}

// value (untempered log-likelihood per chain) and gradient of the tempered log-target of C chains
int ey_mid_eval(ey_plan* pl, const float* theta, const float* temp, int C, float* lik_o, float* grad, hipStream_t s) {
  const EyModel& m = pl->m;
  MidArgs a = {};
  if (!mid_plan(m, a)) EY_FAIL(EY_ERR_UNSUPPORTED, "ey_mid_eval: model not served by the fused mid-size kernel");
  a.theta = theta; a.x = (const float*)m.x; a.y = (const float*)m.y; a.labels = m.labels; a.mu = (const float*)m.mu;
  a.iv = (const float*)m.inv_var; a.temp = temp; a.lik_o = lik_o; a.grad = grad;
  a.C = C; a.N = m.N; a.P = m.P; a.nl = m.nl; a.lik = m.lik;
  a.prior_uniform = pl->prior_uniform ? 1 : 0; a.mu0 = (float)pl->prior_mu0; a.iv0 = (float)pl->prior_iv0;
  for (int l = 0; l <= m.nl; ++l) a.dims[l] = m.dims[l];
  for (int l = 0; l < m.nl; ++l) { a.woff[l] = m.woff[l]; a.boff[l] = m.boff[l]; a.act[l] = m.act[l]; }
  const size_t bytes = (size_t)a.total_floats * sizeof(float);
  const unsigned grid = (unsigned)std::min<int64_t>(C, pl->n_cu > 0 ? pl->n_cu : 256);
  if (m.dims[0] > 32) {
    EY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mid<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(k_mid<true>, dim3(grid), dim3(512), bytes, s, a);
  } else {
    EY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mid<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(k_mid<false>, dim3(grid), dim3(512), bytes, s, a);
  }
  EY_HIP(hipGetLastError());
  return EY_OK;
}
