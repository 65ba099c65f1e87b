// Generic chain-batched kernels: any MLP (dims/bias/activations/likelihood of the plan), f32 and f64.
// One wavefront (64 lanes) per chain; theta, momentum and gradient live in LDS for the whole step.
//
// Reference path restated (paths relative to papamarkou/eeyore):
//   MLP.forward                eeyore/models/mlp.py:45-50
//   BCE-sum / CE-sum           eeyore/constants/constants.py:15-18, eeyore/stats/loss.py:1-11
//   log_lik/log_prior/target   eeyore/models/bayesian_model.py:30-56
//   gradient (autograd there)  eeyore/models/log_target_model.py:15-23
//   HMC leapfrog / draw        eeyore/samplers/hmc.py:100-156
//   MALA draw                  eeyore/samplers/mala.py:46-82
//   MH draw                    eeyore/samplers/metropolis_hastings.py:41-73
//
// Data rows are processed in tiles of 64 (lane <-> row).  Forward is row-parallel (weights broadcast from
// LDS); the weight gradient is parameter-parallel (lane <-> parameter, contraction over the tile's rows);
// the input gradient is row-parallel again.  Tile columns use a stride of 65 elements so that the
// parameter-parallel reads (different rows j, same column n) hit different LDS banks.
#include <atomic>

#include "ey_common.h"

#define TS 65
#define WAVE 64

template <typename T>
struct Num;
template <>
struct Num<float> {
  static __device__ float exp(float v) { return expf(v); }
  static __device__ float log(float v) { return logf(v); }
  static __device__ float tanh(float v) { return tanhf(v); }
  static __device__ float sqrt(float v) { return sqrtf(v); }
};
template <>
struct Num<double> {
  static __device__ double exp(double v) { return ::exp(v); }
  static __device__ double log(double v) { return ::log(v); }
  static __device__ double tanh(double v) { return ::tanh(v); }
  static __device__ double sqrt(double v) { return ::sqrt(v); }
};

template <typename T>
__device__ inline T act_fn(int code, T g) {
  switch (code) {
    case EY_ACT_SIGMOID: return T(1) / (T(1) + Num<T>::exp(-g));
    case EY_ACT_TANH: return Num<T>::tanh(g);
    case EY_ACT_RELU: return g > T(0) ? g : T(0);
    default: return g;
  }
}
template <typename T>
__device__ inline T dact_fn(int code, T h) {
  switch (code) {
    case EY_ACT_SIGMOID: return h * (T(1) - h);
    case EY_ACT_TANH: return T(1) - h * h;
    case EY_ACT_RELU: return h > T(0) ? T(1) : T(0);
    default: return T(1);
  }
}

// A lane-strided pass over n elements with the SAME trip count in every lane: index i = lane + 64 k clamped to n - 1, `on`
// saying whether the lane's element exists.  A lane beyond n in the last round repeats the work of element n - 1's owner --
// the same inputs, hence the same bits to the same address -- and must keep its terms out of sums (select the term's
// INPUT to zero, so that the expression contracts as it did).  No memory operation behind a per-lane branch, no
// loop-carried register under a partial EXEC mask.
#define EY_LANE_PASS(n, i, on)                                                                               \
  for (int ey_k_ = 0, ey_n_ = (n), i = lane < ey_n_ ? lane : ey_n_ - 1, on = lane < ey_n_; ey_k_ < (ey_n_ + WAVE - 1) / WAVE; \
       ++ey_k_, on = lane + ey_k_ * WAVE < ey_n_, i = on ? lane + ey_k_ * WAVE : ey_n_ - 1)

template <typename T>
__device__ inline T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
  return v;
}

// LDS carve for one chain
template <typename T>
struct Lds {
  T* th;   // [P] position
  T* gr;   // [P] gradient at th
  T* a;    // [P] momentum (HMC) / proposal (MALA, MH)
  T* b;    // [P] gradient at the proposal (MALA)
  T* act;  // [hrows * TS] activations of the current row tile, all layers
  T* dl;   // [2 * dmax * TS] delta ping-pong
};

template <typename T>
__device__ inline Lds<T> carve(const EyModel& m, unsigned char* smem, int nvec) {
  Lds<T> l;
  T* p = reinterpret_cast<T*>(smem);
  const int Ppad = (m.P + 3) & ~3;
  l.th = p; p += Ppad;
  l.gr = p; p += Ppad;
  l.a = p; if (nvec > 2) p += Ppad;
  l.b = p; if (nvec > 3) p += Ppad;
  l.act = p; p += m.hrows * TS;
  l.dl = p;
  return l;
}

__host__ __device__ static size_t lds_bytes(const EyModel& m, int nvec, size_t esz) {
  const size_t Ppad = (m.P + 3) & ~3;
  return esz * (nvec * Ppad + (size_t)m.hrows * TS + 2 * (size_t)m.dmax * TS);
}

// ROW WAVES.  A workgroup is one chain.  With the register-resident evaluation (tiny models) and a batch of several 64-row
// tiles, the tiles of one evaluation are independent until the gradient is summed: the workgroup then has up to four
// waves, every wave runs the WHOLE kernel on its own LDS copy of the chain's state (same inputs, same instructions: the
// same bits in every wave -- random draws, proposals, accept decisions), but takes only every NW-th row tile inside an
// evaluation; the waves' partial gradients and log-likelihoods meet in LDS and every wave adds them in the order wave 0,
// 1, 2, 3, so all waves continue with identical totals.  Only wave 0 writes to global memory; the workgroup barriers the
// kernels already have order those writes before the other waves' next reads (one CU, one vector cache).  The number of
// waves is a plan option (EY_OPT_ROW_WAVES): the order in which a gradient is summed differs between one wave and several, so
// a caller who needs a chain's bits to be independent of how many chains share its launch pins it on or off.
#define RW_MAX 4
struct RowWaves {
  int wave, nw;
  void* part;  // exchange buffer [P + 1][nw][64 lanes] of per-lane partial sums, then [P + 4] totals
};
template <typename T>
__device__ inline RowWaves row_waves(const EyModel& m, unsigned char* smem, int nvec) {
  RowWaves rw;
  rw.wave = threadIdx.x >> 6;
  rw.nw = blockDim.x >> 6;
  rw.part = smem + (size_t)rw.nw * lds_bytes(m, nvec, sizeof(T));
  return rw;
}
static size_t row_waves_exchange(const EyModel& m, size_t esz, int nw) {
  return nw > 1 ? esz * ((size_t)(m.P + 1) * nw * WAVE + m.P + 4) : 0;
}
// EY_OPT_ROW_WAVES: off, on (whenever the batch has two row tiles or more), or auto = on while one wave per chain would
// leave SIMDs idle (C <= 4 x CUs: at 256 chains MALA on MLP(2-3-2-1), N = 256, takes 5.4 us per draw instead of 7.7; with
// the chip full of chains the waves' repeated scalar work costs 3 x in throughput, so there it stays off).
static int row_waves_for(const ey_plan* pl, bool tiny, int64_t C) {
  if (!tiny || pl->m.N < 2 * WAVE || pl->row_waves == EY_ROW_WAVES_OFF) return 1;
  if (pl->row_waves == EY_ROW_WAVES_AUTO && C > 4 * (int64_t)(pl->n_cu > 0 ? pl->n_cu : 256)) return 1;
  int nw = std::min(RW_MAX, (pl->m.N + WAVE - 1) / WAVE);
  const size_t esz = pl->dtype == EY_F32 ? 4 : 8;
  while (nw > 1 && row_waves_exchange(pl->m, esz, nw) > 32768) --nw;  // several chains per CU must still fit
  return nw;
}
static size_t lds_total(const EyModel& m, int nvec, size_t esz, int nw) {
  return nw * lds_bytes(m, nvec, esz) + row_waves_exchange(m, esz, nw);
}

// ------------------------------------------------------------------ register-resident evaluation of tiny models
// The models of the reference's own tests and examples are tiny (MLP(2-2-1), (2-3-2-1), (4-3-3): P = 9 .. 27).  The
// tile loop of eval_target below walks such a model through LDS one dependent round trip after the other (a (j, i)
// loop nest of runtime extents, every load behind the previous store): ~20 000 cycles per 64-row tile of MLP(2-3-2-1).
// For at most three layers, at most 8 inputs and every other width at most 4, the same arithmetic fits in registers:
// lane <-> row, the weights as wave-uniform register copies (read once per evaluation from the position in LDS), a row's
// activations and deltas in registers, the weight gradient accumulated PER LANE over the lane's rows and summed over
// the lanes once per evaluation (DPP adds in a fixed order: deterministic).  Every loop is unrolled to its maximum
// extent with wave-uniform guards, so all register arrays are indexed by constants.
#define TINY_D0 8
#define TINY_DH 4
bool ey_generic_tiny_ok(const EyModel& m) {
  if (m.nl < 1 || m.nl > 3 || m.dims[0] > TINY_D0) return false;
  for (int k = 1; k <= m.nl; ++k)
    if (m.dims[k] > TINY_DH) return false;
  return true;
}

// Shape policies of the evaluation below: TinyDyn reads the extents from the model at run time (every unrolled
// iteration behind a wave-uniform guard: ~300 scalar branches per tile of MLP(2-3-2-1)); TinyFix<...> states them at
// compile time, so the guards fold away and only the registers the shape needs remain.  TinyFix exists for the shapes
// the reference's own tests and examples use (mlp.py's default 1-2-1, XOR 2-2-1 and 2-3-2-1, 2-3-3-2, Iris 4-3-3 and
// 4-3-2-3, the banknotes logistic regression 4-1).
struct TinyOff {
  static constexpr bool on = false;
};
struct TinyDyn {
  static constexpr bool on = true;
  static __device__ __forceinline__ int nl(const EyModel& m) { return m.nl; }
  static __device__ __forceinline__ int dim(const EyModel& m, int k) { return m.dims[k]; }
};
template <int NL, int D0, int D1, int D2, int D3>
struct TinyFix {
  static constexpr bool on = true;
  static __device__ __forceinline__ constexpr int nl(const EyModel&) { return NL; }
  static __device__ __forceinline__ constexpr int dim(const EyModel&, int k) { return k == 0 ? D0 : (k == 1 ? D1 : (k == 2 ? D2 : D3)); }
};

template <int CTRL, int ROWMASK, typename T>
__device__ __forceinline__ T tiny_dpp(T v) {
  if constexpr (sizeof(T) == 4) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xF, false));
  } else {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, ROWMASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROWMASK, 0xF, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
  }
}
// wave total as a uniform value: adds inside each 16-lane row, then row_bcast:15 / :31 carry the row totals to lane 63
template <typename T>
__device__ __forceinline__ T tiny_wsum(T v) {
  v += tiny_dpp<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v += tiny_dpp<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v += tiny_dpp<0x141, 0xF>(v);  // row_half_mirror
  v += tiny_dpp<0x140, 0xF>(v);  // row_mirror
  v += tiny_dpp<0x142, 0xA>(v);  // row_bcast:15 (a masked-off row adds the 0 of update_dpp's old value)
  v += tiny_dpp<0x143, 0xC>(v);  // row_bcast:31
  if constexpr (sizeof(T) == 4) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
  } else {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)b, 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
  }
}

// activation of up to four values with the (wave-uniform) switch outside the element loop
template <typename T>
__device__ __forceinline__ void tiny_act(int code, T (&h)[TINY_DH], int n) {
  if (code == EY_ACT_NONE) return;
#pragma unroll
  for (int j = 0; j < TINY_DH; ++j)
    if (j < n) h[j] = act_fn<T>(code, h[j]);
}

// The row loop of eval_target for a tiny model: the sum of the rows' log-likelihood terms (per lane: the caller adds
// the lanes) and, when GRAD, the gradient of the log-likelihood in gr (LDS, canonical layout).
template <typename T, bool GRAD, class S>
__device__ __forceinline__ T tiny_rows(const EyModel& m, const T* th, T* gr, bool has_temp, T temp, T* row_out,
                                       const RowWaves& rw) {
  const int lane = threadIdx.x & (WAVE - 1);
  const T* x = static_cast<const T*>(m.x);
  const T* y = static_cast<const T*>(m.y);
  const int nl = S::nl(m), dK = S::dim(m, nl), d0 = S::dim(m, 0);
  // wave-uniform register copies of the position, padded to [3][4][8 | 4] (+ [3][4] biases)
  T W[3][TINY_DH][TINY_D0], B[3][TINY_DH], G[3][TINY_DH][TINY_D0], GB[3][TINY_DH];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (k < nl) {
      const int din = S::dim(m, k), dout = S::dim(m, k + 1);
#pragma unroll
      for (int j = 0; j < TINY_DH; ++j) {
#pragma unroll
        for (int i = 0; i < (k == 0 ? TINY_D0 : TINY_DH); ++i) {
          W[k][j][i] = (j < dout && i < din) ? th[m.woff[k] + j * din + i] : T(0);
          G[k][j][i] = T(0);
        }
        B[k][j] = (j < dout && m.boff[k] >= 0) ? th[m.boff[k] + j] : T(0);
        GB[k][j] = T(0);
      }
    }
  }
  T lik = T(0);
  for (int n0 = rw.wave * WAVE; n0 < m.N; n0 += rw.nw * WAVE) {
    const int n = n0 + lane;
    const bool valid = n < m.N;
    T h0[TINY_D0], h[3][TINY_DH];  // h[k] = output of layer k
#pragma unroll
    for (int i = 0; i < TINY_D0; ++i) h0[i] = (i < d0 && valid) ? x[(size_t)n * d0 + i] : T(0);
    // ---- forward (mlp.py:45-50)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (k < nl) {
        const int din = S::dim(m, k), dout = S::dim(m, k + 1);
#pragma unroll
        for (int j = 0; j < TINY_DH; ++j) {
          T g = T(0);
          if (j < dout) {
#pragma unroll
            for (int i = 0; i < (k == 0 ? TINY_D0 : TINY_DH); ++i)
              if (i < din) g += (k == 0 ? h0[i] : h[k == 0 ? 0 : k - 1][i]) * W[k][j][i];
            g += B[k][j];  // 0 when the layer has no bias
          }
          h[k][j] = g;
        }
        tiny_act<T>(m.act[k], h[k], dout);
      }
    }
    // ---- likelihood and output delta (constants.py:15-18, loss.py:1-11): the arithmetic of eval_target
    T out[TINY_DH], d[TINY_DH];
#pragma unroll
    for (int j = 0; j < TINY_DH; ++j) {
      out[j] = nl == 1 ? h[0][j] : (nl == 2 ? h[1][j] : h[2][j]);
      d[j] = T(0);
    }
    const int act_out = m.act[nl - 1];
    T row_lik = T(0);
    if (m.lik == EY_LIK_BCE_SUM) {
#pragma unroll
      for (int j = 0; j < TINY_DH; ++j)
        if (j < dK) {
          const T o = out[j];
          const T yy = valid ? y[(size_t)n * dK + j] : T(0);
          const T term = Num<T>::log(o) * yy + Num<T>::log(T(1) - o) * (T(1) - yy);
          lik += valid ? term : T(0);
          row_lik += term;
          if (GRAD) {
            const T dd = (yy / o - (T(1) - yy) / (T(1) - o)) * dact_fn<T>(act_out, o);
            d[j] = valid ? dd : T(0);
          }
        }
    } else {
      const int lab = valid ? m.labels[n] : 0;
      T mx = out[0];
#pragma unroll
      for (int j = 1; j < TINY_DH; ++j)
        if (j < dK) mx = fmax(mx, out[j]);
      T ssum = T(0), olab = out[0];
#pragma unroll
      for (int j = 0; j < TINY_DH; ++j)
        if (j < dK) {
          ssum += Num<T>::exp(out[j] - mx);
          if (j == lab) olab = out[j];
        }
      row_lik = olab - (mx + Num<T>::log(ssum));
      lik += valid ? row_lik : T(0);
      if (GRAD) {
#pragma unroll
        for (int j = 0; j < TINY_DH; ++j)
          if (j < dK) {
            const T dd = ((j == lab ? T(1) : T(0)) - Num<T>::exp(out[j] - mx) / ssum) * dact_fn<T>(act_out, out[j]);
            d[j] = valid ? dd : T(0);
          }
      }
    }
    if (row_out && valid) row_out[n] = has_temp ? row_lik * temp : row_lik;
    if (GRAD) {
      // ---- backward: dW_k += delta_k (x) h_{k-1}, db_k += delta_k, delta_{k-1} = (W_k^T delta_k) * act'(h_{k-1})
#pragma unroll
      for (int k = 2; k >= 0; --k) {
        if (k < nl) {
          const int din = S::dim(m, k), dout = S::dim(m, k + 1);
#pragma unroll
          for (int j = 0; j < TINY_DH; ++j)
            if (j < dout) {
#pragma unroll
              for (int i = 0; i < (k == 0 ? TINY_D0 : TINY_DH); ++i)
                if (i < din) G[k][j][i] += d[j] * (k == 0 ? h0[i] : h[k == 0 ? 0 : k - 1][i]);
              GB[k][j] += d[j];
            }
          if (k > 0) {
            T dn[TINY_DH];
#pragma unroll
            for (int i = 0; i < TINY_DH; ++i) {
              T a = T(0);
              if (i < din) {
#pragma unroll
                for (int j = 0; j < TINY_DH; ++j)
                  if (j < dout) a += d[j] * W[k][j][i];
                a *= dact_fn<T>(m.act[k - 1], h[k - 1][i]);
              }
              dn[i] = a;
            }
#pragma unroll
            for (int i = 0; i < TINY_DH; ++i) d[i] = dn[i];
          }
        }
      }
    }
  }
  if (rw.nw > 1) {
    // ---- ROW WAVES: every wave leaves its per-lane partial sums in the exchange buffer; the parameters (and the
    // log-likelihood, as entry P) are then shared out over the waves, each adding the waves' partials lane by lane in the
    // order wave 0, 1, ... and reducing over the lanes as below; every wave copies the totals into its own gradient.
    T* ex = static_cast<T*>(rw.part);
    T* totals = ex + (size_t)(m.P + 1) * rw.nw * WAVE;
    auto slot = [&](int idx) { return ex + ((size_t)idx * rw.nw + rw.wave) * WAVE + lane; };
    if (GRAD) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (k < nl) {
          const int din = S::dim(m, k), dout = S::dim(m, k + 1);
#pragma unroll
          for (int j = 0; j < TINY_DH; ++j)
            if (j < dout) {
#pragma unroll
              for (int i = 0; i < (k == 0 ? TINY_D0 : TINY_DH); ++i)
                if (i < din) *slot(m.woff[k] + j * din + i) = G[k][j][i];
              if (m.boff[k] >= 0) *slot(m.boff[k] + j) = GB[k][j];
            }
        }
      }
    }
    *slot(m.P) = lik;
    __syncthreads();
    for (int idx = (GRAD ? rw.wave : m.P + rw.wave); idx <= m.P; idx += rw.nw) {
      const T* src = ex + (size_t)idx * rw.nw * WAVE + lane;
      T v = src[0];
      for (int w = 1; w < rw.nw; ++w) v += src[w * WAVE];
      const T tot = tiny_wsum<T>(v);
      totals[idx] = tot;  // (wave-uniform: every lane stores it)
    }
    __syncthreads();
    if (GRAD)
      for (int i = lane; i < m.P; i += WAVE) gr[i] = totals[i];
    return lane == 0 ? totals[m.P] : T(0);  // the caller's wave_sum hands it to every lane
  }
  if (GRAD) {
    // ---- the lanes' partial gradients, summed in a fixed order, into the canonical layout
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (k < nl) {
        const int din = S::dim(m, k), dout = S::dim(m, k + 1);
#pragma unroll
        for (int j = 0; j < TINY_DH; ++j)
          if (j < dout) {
#pragma unroll
            for (int i = 0; i < (k == 0 ? TINY_D0 : TINY_DH); ++i)
              if (i < din) {
                // (the total is wave-uniform: every lane stores it -- no store behind a per-lane branch while the
                // partial gradients of all the other elements are live in registers)
                const T tot = tiny_wsum<T>(G[k][j][i]);
                gr[m.woff[k] + j * din + i] = tot;
              }
            if (m.boff[k] >= 0) {
              const T tot = tiny_wsum<T>(GB[k][j]);
              gr[m.boff[k] + j] = tot;
            }
          }
      }
    }
  }
  return lik;
}

// log-target (and gradient when GRAD) of the position in `th`; result broadcast to every lane.
// gr receives the gradient of the (tempered) log-target.  lik/prior are the tempered parts.
// the chain's N(0,1) stream for elements 0..P-1 into an LDS array, one block of four per lane and round
template <typename T>
__device__ inline void fill_normals(T* dst, const EyRng& rn, int P) {
  for (int b = threadIdx.x & (WAVE - 1); 4 * b < P; b += WAVE) {
    T o[4];
    ey_rng_normal4<T>(rn, (uint32_t)b, o);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (4 * b + j < P) dst[4 * b + j] = o[j];
  }
  __syncthreads();
}

template <typename T, bool GRAD, class TINY = TinyOff>
__device__ T eval_target(const EyModel& m, const Lds<T>& l, const T* th, T* gr, bool has_temp, T temp, T* lik_out,
                         T* prior_out, T* row_out = nullptr, const RowWaves rw = RowWaves{0, 1, nullptr}) {
  const int lane = threadIdx.x & (WAVE - 1);
  const T* x = static_cast<const T*>(m.x);
  const T* y = static_cast<const T*>(m.y);
  const int nl = m.nl;
  const int dK = m.dims[nl];
  T lik = T(0);
  if constexpr (TINY::on) {
    __syncthreads();  // the position written by the caller is visible
    lik = tiny_rows<T, GRAD, TINY>(m, th, gr, has_temp, temp, row_out, rw);
  } else {
  if (GRAD) {
    for (int i = lane; i < m.P; i += WAVE) gr[i] = T(0);
  }
  for (int n0 = 0; n0 < m.N; n0 += WAVE) {
    const int rows = min(WAVE, m.N - n0);
    const int n = n0 + lane;
    const bool valid = lane < rows;
    __syncthreads();  // previous tile's parameter-parallel reads are done; th/gr initialisation visible
    for (int i = 0; i < m.dims[0]; ++i) l.act[(m.hoff[0] + i) * TS + lane] = valid ? x[(size_t)n * m.dims[0] + i] : T(0);
    // ---- forward, row-parallel (mlp.py:45-50)
    for (int k = 0; k < nl; ++k) {
      const int din = m.dims[k], dout = m.dims[k + 1];
      const T* W = th + m.woff[k];
      const T* hin = l.act + m.hoff[k] * TS + lane;
      T* hout = l.act + m.hoff[k + 1] * TS + lane;
      // four outputs at a time: one read of the row's input feeds four sums and nothing is stored inside the i loop,
      // so its loads pipeline (one output at a time every load waited behind the previous output's store); each sum
      // still adds its terms in the order i = 0, 1, ..., so the results are those of the plain loop, bit for bit
      for (int j0 = 0; j0 < dout; j0 += 4) {
        const int nj = min(4, dout - j0);
        const T* w0 = W + j0 * din;
        const T* w1 = w0 + (nj > 1 ? din : 0);  // rows beyond dout alias row j0: read, never stored
        const T* w2 = w0 + (nj > 2 ? 2 * din : 0);
        const T* w3 = w0 + (nj > 3 ? 3 * din : 0);
        T g0 = T(0), g1 = T(0), g2 = T(0), g3 = T(0);
#pragma unroll 4
        for (int i = 0; i < din; ++i) {
          const T h = hin[i * TS];
          g0 += h * w0[i];
          g1 += h * w1[i];
          g2 += h * w2[i];
          g3 += h * w3[i];
        }
        auto put = [&](int q, T g) {
          if (q < nj) {
            if (m.boff[k] >= 0) g += th[m.boff[k] + j0 + q];
            hout[(j0 + q) * TS] = act_fn<T>(m.act[k], g);
          }
        };
        put(0, g0); put(1, g1); put(2, g2); put(3, g3);
      }
    }
    // ---- likelihood and output delta
    const T* out = l.act + m.hoff[nl] * TS + lane;
    T* dcur = l.dl;
    T* dnext = l.dl + m.dmax * TS;
    T row_lik = T(0);  // this row's term of the log-likelihood sum (ey_log_lik_rows)
    if (m.lik == EY_LIK_BCE_SUM) {
      for (int j = 0; j < dK; ++j) {
        const T o = out[j * TS];
        const T yy = valid ? y[(size_t)n * dK + j] : T(0);
        // naive logs exactly as eeyore/stats/loss.py:2 (NaN once a sigmoid saturates)
        const T term = Num<T>::log(o) * yy + Num<T>::log(T(1) - o) * (T(1) - yy);
        if (valid) lik += term;
        row_lik += term;
        if (GRAD) {
          const T d = (yy / o - (T(1) - yy) / (T(1) - o)) * dact_fn<T>(m.act[nl - 1], o);
          dcur[j * TS + lane] = valid ? d : T(0);
        }
      }
    } else {
      const int lab = valid ? m.labels[n] : 0;
      T mx = out[0];
      for (int j = 1; j < dK; ++j) mx = fmax(mx, out[j * TS]);
      T ssum = T(0);
      for (int j = 0; j < dK; ++j) ssum += Num<T>::exp(out[j * TS] - mx);
      row_lik = out[lab * TS] - (mx + Num<T>::log(ssum));
      if (valid) lik += row_lik;
      if (GRAD) {
        for (int j = 0; j < dK; ++j) {
          const T o = out[j * TS];
          const T d = ((j == lab ? T(1) : T(0)) - Num<T>::exp(o - mx) / ssum) * dact_fn<T>(m.act[nl - 1], o);
          dcur[j * TS + lane] = valid ? d : T(0);
        }
      }
    }
    if (row_out && valid) row_out[n] = has_temp ? row_lik * temp : row_lik;
    if (GRAD) {
      // ---- backward
      for (int k = nl - 1; k >= 0; --k) {
        const int din = m.dims[k], dout = m.dims[k + 1];
        __syncthreads();  // delta_k of every row visible
        // dW_k[j][i] += sum_n delta[j][n] * h_{k}[i][n]   (parameter-parallel)
        const T* hin_t = l.act + m.hoff[k] * TS;
        for (int idx = lane; idx < dout * din; idx += WAVE) {
          const int j = idx / din, i = idx - j * din;
          const T* dj = dcur + j * TS;
          const T* hi = hin_t + i * TS;
          T acc = T(0);
          // few lanes are active when a layer is small: keep several LDS reads in flight per lane
#pragma unroll 8
          for (int r = 0; r < rows; ++r) acc += dj[r] * hi[r];
          gr[m.woff[k] + idx] += acc;
        }
        if (m.boff[k] >= 0) {
          for (int j = lane; j < dout; j += WAVE) {
            const T* dj = dcur + j * TS;
            T acc = T(0);
#pragma unroll 8
            for (int r = 0; r < rows; ++r) acc += dj[r];
            gr[m.boff[k] + j] += acc;
          }
        }
        if (k > 0) {
          // delta_{k-1}[i][n] = (sum_j delta_k[j][n] W_k[j][i]) * act'(h_k[i][n])   (row-parallel)
          const T* W = th + m.woff[k];
          for (int i0 = 0; i0 < din; i0 += 4) {  // four inputs at a time, each sum in the order j = 0, 1, ...
            const int ni = min(4, din - i0);
            const int o1 = ni > 1 ? 1 : 0, o2 = ni > 2 ? 2 : 0, o3 = ni > 3 ? 3 : 0;
            T a0 = T(0), a1 = T(0), a2 = T(0), a3 = T(0);
#pragma unroll 4
            for (int j = 0; j < dout; ++j) {
              const T d = dcur[j * TS + lane];
              const T* w = W + j * din + i0;
              a0 += d * w[0];
              a1 += d * w[o1];
              a2 += d * w[o2];
              a3 += d * w[o3];
            }
            auto put = [&](int q, T a) {
              if (q < ni) dnext[(i0 + q) * TS + lane] = a * dact_fn<T>(m.act[k - 1], hin_t[(i0 + q) * TS + lane]);
            };
            put(0, a0); put(1, a1); put(2, a2); put(3, a3);
          }
          T* t = dcur; dcur = dnext; dnext = t;
        }
      }
    }
  }
  }  // !TINY
  __syncthreads();
  lik = wave_sum(lik);
  // ---- prior (bayesian_model.py:46-50), elementwise Normal(mu, sigma)
  const T* mu = static_cast<const T*>(m.mu);
  const T* iv = static_cast<const T*>(m.inv_var);
  T q = T(0);
  // (EY_LANE_PASS: the same trip count in every lane -- a sum carried across a loop whose last round runs under a partial
  // EXEC mask is what the fast-allocator build of this unit got wrong)
  EY_LANE_PASS(m.P, i, on) {
    const T d = th[i] - mu[i];
    const T dz = on ? d : T(0);
    q += dz * dz * iv[i];
    if (GRAD) {
      T g = gr[i] - d * iv[i];
      if (has_temp) g *= temp;
      gr[i] = g;
    }
  }
  q = wave_sum(q);
  T prior = T(m.prior_const) - T(0.5) * q;
  if (has_temp) { lik *= temp; prior *= temp; }
  if (lik_out) *lik_out = lik;
  if (prior_out) *prior_out = prior;
  __syncthreads();
  return lik + prior;
}

// ----------------------------------------------------------------------------------------------- kernels
template <typename T, bool GRAD, class TINY, bool RW = false>
__global__ void __launch_bounds__(RW ? RW_MAX * WAVE : WAVE) k_log_target(EyModel m, const T* theta, const T* temp, T* lik_o, T* prior_o,
                                                     T* target_o, T* grad_o, T* rows_o) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const RowWaves rw = RW ? row_waves<T>(m, smem, 2) : RowWaves{0, 1, nullptr};  // RW false: one wave, all of this folds away
  const Lds<T> l = carve<T>(m, smem + (RW ? (size_t)rw.wave * lds_bytes(m, 2, sizeof(T)) : 0), 2);
  const bool w0 = rw.wave == 0;  // the wave that writes to global memory (ROW WAVES)
  const int64_t c = blockIdx.x;
  const int lane = threadIdx.x & (WAVE - 1);
  for (int i = lane; i < m.P; i += WAVE) l.th[i] = theta[c * m.P + i];
  const bool ht = temp != nullptr;
  const T tc = ht ? temp[c] : T(1);
  T lik, prior;
  const T t = eval_target<T, GRAD, TINY>(m, l, l.th, l.gr, ht, tc, &lik, &prior, rows_o ? rows_o + c * m.N : nullptr, rw);
  if (lane == 0 && w0) {
    if (lik_o) lik_o[c] = lik;
    if (prior_o) prior_o[c] = prior;
    if (target_o) target_o[c] = t;
  }
  if (GRAD && w0) {
    for (int i = lane; i < m.P; i += WAVE) grad_o[c * m.P + i] = l.gr[i];
  }
}

template <typename T, class TINY, bool RW = false>
__global__ void __launch_bounds__(RW ? RW_MAX * WAVE : WAVE) k_hmc(EyModel m, T* theta, T* target, T* grad, const T* p0, const T* u_in,
                                              T step, const T* step_vec, int L, const T* temp, uint64_t seed,
                                              uint64_t iter0, uint64_t chain_offset, int recompute,
                                              unsigned char* accepted, T* rate_o, T* hcur_o, T* hprop_o, int n_iters,
                                              T* rec_samples, T* rec_targets, unsigned char* rec_accepted,
                                              int* accept_count, int64_t C) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const RowWaves rw = RW ? row_waves<T>(m, smem, 3) : RowWaves{0, 1, nullptr};  // RW false: one wave, all of this folds away
  const Lds<T> l = carve<T>(m, smem + (RW ? (size_t)rw.wave * lds_bytes(m, 3, sizeof(T)) : 0), 3);
  const bool w0 = rw.wave == 0;  // the wave that writes to global memory (ROW WAVES)
  const int64_t c = blockIdx.x;
  const int lane = RW ? (threadIdx.x & (WAVE - 1)) : threadIdx.x;
  const int P = m.P;
  const bool ht = temp != nullptr;
  const T tc = ht ? temp[c] : T(1);
  const T eps = step_vec ? step_vec[c] : step;
  T* p = l.a;
  T t_state = target[c];
  // ey_hmc_run: n_iters draws in one launch; every lane re-reads only what it wrote itself (theta, grad), the
  // log-target is carried in a register
  for (int it = 0; it < n_iters; ++it) {
  const uint64_t iter = iter0 + (uint64_t)it;
  // momentum ~ N(0, I)  (hmc.py:134)
  const EyRng rn = ey_rng_make(seed, chain_offset + (uint64_t)c, iter, EY_STREAM_NORMAL);
  T kin = T(0);
  if (!p0) fill_normals<T>(p, rn, P);
  EY_LANE_PASS(P, i, on) {
    l.th[i] = theta[c * P + i];
    l.gr[i] = grad[c * P + i];
    const T pi = p0 ? p0[c * P + i] : p[i];
    p[i] = pi;
    const T pz = on ? pi : T(0);
    kin += pz * pz;
  }
  kin = wave_sum(kin);
  const T t_cur = t_state;
  const T h_cur = -t_cur + T(0.5) * kin;  // hmc.py:91-98,137
  __syncthreads();
  T t = t_cur;
  if (recompute) t = eval_target<T, true, TINY>(m, l, l.th, l.gr, ht, tc, nullptr, nullptr, nullptr, rw);  // hmc.py:104
  // leapfrog (hmc.py:100-124); grad_potential = -grad
  for (int i = lane; i < P; i += WAVE) p[i] = p[i] + T(0.5) * eps * l.gr[i];
  for (int k = 1; k <= L; ++k) {
    for (int i = lane; i < P; i += WAVE) l.th[i] = l.th[i] + eps * p[i];
    t = eval_target<T, true, TINY>(m, l, l.th, l.gr, ht, tc, nullptr, nullptr, nullptr, rw);
    const T w = (k < L) ? eps : T(0.5) * eps;
    for (int i = lane; i < P; i += WAVE) p[i] = p[i] + w * l.gr[i];
  }
  kin = T(0);
  EY_LANE_PASS(P, i, on) {  // p -> -p leaves it unchanged (hmc.py:122)
    const T pz = on ? p[i] : T(0);
    kin += pz * pz;
  }
  kin = wave_sum(kin);
  const T h_prop = -t + T(0.5) * kin;
  T rate = Num<T>::exp(h_cur - h_prop);  // hmc.py:143-146
  if (rate > T(1)) rate = T(1);
  const EyRng ru = ey_rng_make(seed, chain_offset + (uint64_t)c, iter, EY_STREAM_UNIFORM);
  const T u = u_in ? u_in[c] : ey_rng_uniform<T>(ru);
  const bool acc = u < rate;  // strict <; NaN rate => reject (hmc.py:148)
  if (acc && w0) {
    for (int i = lane; i < P; i += WAVE) {
      theta[c * P + i] = l.th[i];
      grad[c * P + i] = l.gr[i];
    }
  }
  if (acc) t_state = t;
  if (rec_samples && w0) {  // the state the chain is left in (what ChainList.update stores, chain_list.py:64-67)
    T* so = rec_samples + ((int64_t)it * C + c) * P;
    for (int i = lane; i < P; i += WAVE) so[i] = acc ? l.th[i] : theta[c * P + i];
  }
  if (lane == 0 && w0) {
    if (acc) target[c] = t;
    accepted[c] = acc ? 1 : 0;
    if (rate_o) rate_o[c] = rate;
    if (hcur_o) hcur_o[c] = h_cur;
    if (hprop_o) hprop_o[c] = h_prop;
    if (rec_targets) rec_targets[(int64_t)it * C + c] = t_state;
    if (rec_accepted) rec_accepted[(int64_t)it * C + c] = acc ? 1 : 0;
    if (accept_count && acc) accept_count[c] += 1;
  }
  __syncthreads();
  }
}

// HMC.leapfrog as a standalone operator (hmc.py:100-124): L+1 evaluations, momentum negated.
template <typename T, class TINY, bool RW = false>
__global__ void __launch_bounds__(RW ? RW_MAX * WAVE : WAVE) k_leapfrog(EyModel m, T* theta, T* pio, T step, const T* step_vec, int L,
                                                   const T* temp, T* target, T* grad) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const RowWaves rw = RW ? row_waves<T>(m, smem, 3) : RowWaves{0, 1, nullptr};  // RW false: one wave, all of this folds away
  const Lds<T> l = carve<T>(m, smem + (RW ? (size_t)rw.wave * lds_bytes(m, 3, sizeof(T)) : 0), 3);
  const bool w0 = rw.wave == 0;  // the wave that writes to global memory (ROW WAVES)
  const int64_t c = blockIdx.x;
  const int lane = RW ? (threadIdx.x & (WAVE - 1)) : threadIdx.x;
  const int P = m.P;
  const bool ht = temp != nullptr;
  const T tc = ht ? temp[c] : T(1);
  const T eps = step_vec ? step_vec[c] : step;
  T* p = l.a;
  for (int i = lane; i < P; i += WAVE) {
    l.th[i] = theta[c * P + i];
    p[i] = pio[c * P + i];
  }
  __syncthreads();
  T t = eval_target<T, true, TINY>(m, l, l.th, l.gr, ht, tc, nullptr, nullptr, nullptr, rw);
  for (int i = lane; i < P; i += WAVE) p[i] = p[i] + T(0.5) * eps * l.gr[i];
  for (int k = 1; k <= L; ++k) {
    for (int i = lane; i < P; i += WAVE) l.th[i] = l.th[i] + eps * p[i];
    t = eval_target<T, true, TINY>(m, l, l.th, l.gr, ht, tc, nullptr, nullptr, nullptr, rw);
    const T w = (k < L) ? eps : T(0.5) * eps;
    for (int i = lane; i < P; i += WAVE) p[i] = p[i] + w * l.gr[i];
  }
  if (w0) {
    for (int i = lane; i < P; i += WAVE) {
      theta[c * P + i] = l.th[i];
      pio[c * P + i] = -p[i];
      grad[c * P + i] = l.gr[i];
    }
    if (lane == 0) target[c] = t;
  }
}

template <typename T, class TINY, bool RW = false>
__global__ void __launch_bounds__(RW ? RW_MAX * WAVE : WAVE) k_mala(EyModel m, T* theta, T* target, T* grad, const T* z_in, const T* u_in,
                                               T step, T sqrt_step, const T* step_vec, const T* temp, uint64_t seed,
                                               uint64_t iter0, uint64_t chain_offset, unsigned char* accepted,
                                               T* log_rate_o, EyRun run, int64_t C) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const RowWaves rw = RW ? row_waves<T>(m, smem, 4) : RowWaves{0, 1, nullptr};  // RW false: one wave, all of this folds away
  const Lds<T> l = carve<T>(m, smem + (RW ? (size_t)rw.wave * lds_bytes(m, 4, sizeof(T)) : 0), 4);
  const bool w0 = rw.wave == 0;  // the wave that writes to global memory (ROW WAVES)
  const int64_t c = blockIdx.x;
  const int lane = RW ? (threadIdx.x & (WAVE - 1)) : threadIdx.x;
  const int P = m.P;
  const bool ht = temp != nullptr;
  const T tc = ht ? temp[c] : T(1);
  const T eps = step_vec ? step_vec[c] : step;
  const T sc = step_vec ? Num<T>::sqrt(eps) : sqrt_step;  // scale = sqrt(step) (mala.py:39)
  const T inv2v = T(1) / (T(2) * sc * sc);
  T* prop = l.a;
  T* gp = l.b;
  T t_state = target[c];
  // ey_mala_run: n_iters draws in one launch; every lane re-reads only what it wrote itself (theta, grad), the
  // log-target is carried in a register
  for (int it = 0; it < run.n_iters; ++it) {
  const uint64_t iter = iter0 + (uint64_t)it;
  const EyRng rn = ey_rng_make(seed, chain_offset + (uint64_t)c, iter, EY_STREAM_NORMAL);
  T qf = T(0);
  if (!z_in) fill_normals<T>(prop, rn, P);
  EY_LANE_PASS(P, i, on) {
    const T th = theta[c * P + i], g = grad[c * P + i];
    l.th[i] = th;
    l.gr[i] = g;
    const T zi = z_in ? z_in[c * P + i] : prop[i];
    const T loc = th + T(0.5) * eps * g;  // kernel_mean (mala.py:35-36)
    const T pr = loc + sc * zi;           // Normal(loc, scale).sample()
    prop[i] = pr;
    const T d = on ? pr - loc : T(0);
    qf += d * d;
  }
  __syncthreads();
  const T tv = eval_target<T, true, TINY>(m, l, prop, gp, ht, tc, nullptr, nullptr, nullptr, rw);
  T qb = T(0);
  EY_LANE_PASS(P, i, on) {
    const T loc2 = prop[i] + T(0.5) * eps * gp[i];
    const T d = on ? l.th[i] - loc2 : T(0);
    qb += d * d;
  }
  qf = wave_sum(qf);
  qb = wave_sum(qb);
  // log q terms share -P log(scale) - P/2 log(2 pi): they cancel in log_rate (mala.py:58-64)
  const T log_rate = (tv - t_state) + qf * inv2v - qb * inv2v;
  const EyRng ru = ey_rng_make(seed, chain_offset + (uint64_t)c, iter, EY_STREAM_UNIFORM);
  const T u = u_in ? u_in[c] : ey_rng_uniform<T>(ru);
  const bool acc = Num<T>::log(u) < log_rate;  // mala.py:66
  if (acc) {
    t_state = tv;
    if (w0)
      for (int i = lane; i < P; i += WAVE) {
        theta[c * P + i] = prop[i];
        grad[c * P + i] = gp[i];
      }
  }
  if (run.samples && w0) {  // the state the chain is left in (what ChainList.update stores, chain_list.py:64-67)
    T* so = static_cast<T*>(run.samples) + ((int64_t)it * C + c) * P;
    for (int i = lane; i < P; i += WAVE) so[i] = acc ? prop[i] : l.th[i];
  }
  if (lane == 0 && w0) {
    if (acc) target[c] = tv;
    accepted[c] = acc ? 1 : 0;
    if (log_rate_o) log_rate_o[c] = log_rate;
    if (run.targets) static_cast<T*>(run.targets)[(int64_t)it * C + c] = t_state;
    if (run.accepted) static_cast<unsigned char*>(run.accepted)[(int64_t)it * C + c] = acc ? 1 : 0;
    if (run.accept_count && acc) run.accept_count[c] += 1;
  }
  __syncthreads();
  }
}

template <typename T, class TINY, bool RW = false>
__global__ void __launch_bounds__(RW ? RW_MAX * WAVE : WAVE) k_mh(EyModel m, T* theta, T* target, const T* z_in, const T* u_in,
                                             const T* scale, const T* temp, uint64_t seed, uint64_t iter0,
                                             uint64_t chain_offset, unsigned char* accepted, T* log_rate_o, EyRun run,
                                             int64_t C) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const RowWaves rw = RW ? row_waves<T>(m, smem, 2) : RowWaves{0, 1, nullptr};  // RW false: one wave, all of this folds away
  const Lds<T> l = carve<T>(m, smem + (RW ? (size_t)rw.wave * lds_bytes(m, 2, sizeof(T)) : 0), 2);
  const bool w0 = rw.wave == 0;  // the wave that writes to global memory (ROW WAVES)
  const int64_t c = blockIdx.x;
  const int lane = RW ? (threadIdx.x & (WAVE - 1)) : threadIdx.x;
  const int P = m.P;
  const bool ht = temp != nullptr;
  const T tc = ht ? temp[c] : T(1);
  T t_state = target[c];
  for (int it = 0; it < run.n_iters; ++it) {  // ey_mh_run: see k_mala
  const uint64_t iter = iter0 + (uint64_t)it;
  const EyRng rn = ey_rng_make(seed, chain_offset + (uint64_t)c, iter, EY_STREAM_NORMAL);
  if (!z_in) fill_normals<T>(l.th, rn, P);
  for (int i = lane; i < P; i += WAVE) {
    const T zi = z_in ? z_in[c * P + i] : l.th[i];
    l.th[i] = theta[c * P + i] + scale[i] * zi;  // NormalKernel(theta, scale).sample()
  }
  __syncthreads();
  const T tv = eval_target<T, false, TINY>(m, l, l.th, l.gr, ht, tc, nullptr, nullptr, nullptr, rw);
  const T log_rate = tv - t_state;  // symmetric kernel (metropolis_hastings.py:50)
  const EyRng ru = ey_rng_make(seed, chain_offset + (uint64_t)c, iter, EY_STREAM_UNIFORM);
  const T u = u_in ? u_in[c] : ey_rng_uniform<T>(ru);
  const bool acc = Num<T>::log(u) < log_rate;  // :56
  if (acc) {
    t_state = tv;
    if (w0)
      for (int i = lane; i < P; i += WAVE) theta[c * P + i] = l.th[i];
  }
  if (run.samples && w0) {
    T* so = static_cast<T*>(run.samples) + ((int64_t)it * C + c) * P;
    for (int i = lane; i < P; i += WAVE) so[i] = acc ? l.th[i] : theta[c * P + i];
  }
  if (lane == 0 && w0) {
    if (acc) target[c] = tv;
    accepted[c] = acc ? 1 : 0;
    if (log_rate_o) log_rate_o[c] = log_rate;
    if (run.targets) static_cast<T*>(run.targets)[(int64_t)it * C + c] = t_state;
    if (run.accepted) static_cast<unsigned char*>(run.accepted)[(int64_t)it * C + c] = acc ? 1 : 0;
    if (run.accept_count && acc) run.accept_count[c] += 1;
  }
  __syncthreads();
  }
}

// ----------------------------------------------------------------------------------------------- host launchers

// Where the register-resident evaluation pays (tools/tiny_scan.py, profiles/r02_tiny_scan.txt): it has a fixed cost per
// evaluation (the register copies of the position, one wave reduction per parameter), so it is taken from two 64-row
// tiles up: 1.6-2.8 x the LDS loop at a few hundred chains, where that loop's dependent round trips are exposed, 1.1-1.7 x
// in f32 when the chains fill the chip; in f64 (one wave per SIMD) a chip full of chains is 0.75-1.1 x.  The rule looks
// at the batch only, not at the number of chains: the arithmetic a chain sees must not depend on how many chains (or
// GPUs) run beside it.  ey_debug_set_variant bit 8: never (A/B, tests), bit 9: whenever the model qualifies.
// 0: the LDS tile loop, 1: TinyDyn, 2..: the compile-time shapes (taken for any batch: their fixed cost is a handful of
// loads and reductions)
static int tiny_kind(const ey_plan* pl) {
  const EyModel& m = pl->m;
  if (!ey_generic_tiny_ok(m) || EY_VBIT(8)) return 0;
  auto is = [&](int nl, int a, int b, int c, int d) {
    return m.nl == nl && m.dims[0] == a && m.dims[1] == b && (nl < 2 || m.dims[2] == c) && (nl < 3 || m.dims[3] == d);
  };
  if (is(2, 2, 2, 1, 0)) return 2;
  if (is(3, 2, 3, 2, 1)) return 3;
  if (is(2, 4, 3, 3, 0)) return 4;
  if (is(3, 4, 3, 2, 3)) return 5;
  if (is(2, 1, 2, 1, 0)) return 6;
  if (is(3, 2, 3, 3, 2)) return 7;
  if (is(1, 4, 1, 0, 0)) return 8;
  return (EY_VBIT(9) || m.N >= 128) ? 1 : 0;
}
template <typename F>
static int tiny_dispatch(const ey_plan* pl, F f) {
  switch (tiny_kind(pl)) {
    case 1: return f(TinyDyn{});
    case 2: return f(TinyFix<2, 2, 2, 1, 0>{});
    case 3: return f(TinyFix<3, 2, 3, 2, 1>{});
    case 4: return f(TinyFix<2, 4, 3, 3, 0>{});
    case 5: return f(TinyFix<3, 4, 3, 2, 3>{});
    case 6: return f(TinyFix<2, 1, 2, 1, 0>{});
    case 7: return f(TinyFix<3, 2, 3, 3, 2>{});
    case 8: return f(TinyFix<1, 4, 1, 0, 0>{});
    default: return f(TinyOff{});
  }
}
#define EY_TINY_DISPATCH(fn, ...)                                                                      \
  tiny_dispatch(pl, [&](auto tiny_tag) {                                                               \
    typedef decltype(tiny_tag) TinyS;                                                                  \
    return pl->dtype == EY_F32 ? fn<float, TinyS>(__VA_ARGS__) : fn<double, TinyS>(__VA_ARGS__);       \
  })

template <typename K>
static int prep(K kernel, size_t bytes) {
  if (bytes > 160 * 1024) EY_FAIL(EY_ERR_UNSUPPORTED, "generic kernel: model does not fit the 160 KiB LDS of a CU");
  if (bytes > 48 * 1024)
    EY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)bytes));
  return EY_OK;
}

// the kernels' row-waves instantiation (RW) where the launch uses several waves per chain, the lean one otherwise
template <class TINY, typename F>
static int with_row_waves(int nw, F f) {
  if constexpr (TINY::on) {
    if (nw > 1) return f(std::true_type{});
  }
  return f(std::false_type{});
}

template <typename T, class TINY>
static int launch_log_target(ey_plan* pl, const void* theta, const void* temp, int64_t C, void* lik, void* prior,
                             void* target, void* grad, hipStream_t s, void* rows = nullptr) {
  const int nw = row_waves_for(pl, TINY::on, C);
  const size_t bytes = lds_total(pl->m, 2, sizeof(T), nw);
  return with_row_waves<TINY>(nw, [&](auto rwtag) -> int {
    constexpr bool RW = decltype(rwtag)::value;
    int rc;
    if (grad) {
      if ((rc = prep(k_log_target<T, true, TINY, RW>, bytes))) return rc;
      hipLaunchKernelGGL((k_log_target<T, true, TINY, RW>), dim3((unsigned)C), dim3(nw * WAVE), bytes, s, pl->m,
                         (const T*)theta, (const T*)temp, (T*)lik, (T*)prior, (T*)target, (T*)grad, (T*)nullptr);
    } else {
      if ((rc = prep(k_log_target<T, false, TINY, RW>, bytes))) return rc;
      hipLaunchKernelGGL((k_log_target<T, false, TINY, RW>), dim3((unsigned)C), dim3(nw * WAVE), bytes, s, pl->m,
                         (const T*)theta, (const T*)temp, (T*)lik, (T*)prior, (T*)target, (T*)nullptr, (T*)rows);
    }
    EY_HIP(hipGetLastError());
    return (int)EY_OK;
  });
}

int ey_generic_log_target(ey_plan* pl, const void* theta, const void* temp, int64_t C, void* lik, void* prior,
                          void* target, void* grad, hipStream_t s) {
  return EY_TINY_DISPATCH(launch_log_target, pl, theta, temp, C, lik, prior, target, grad, s);
}

int ey_generic_log_lik_rows(ey_plan* pl, const void* theta, const void* temp, int64_t C, void* rows, hipStream_t s) {
  return EY_TINY_DISPATCH(launch_log_target, pl, theta, temp, C, nullptr, nullptr, nullptr, nullptr, s, rows);
}

template <typename T, class TINY>
static int launch_hmc(ey_plan* pl, void* theta, void* target, void* grad, const void* p0, const void* u, double step,
                      const void* step_vec, int L, const void* temp, int64_t C, uint64_t seed, uint64_t iter,
                      uint64_t chain_offset, uint32_t flags, void* accepted, void* rate, void* hcur, void* hprop,
                      hipStream_t s, const EyRun* run) {
  const int nw = row_waves_for(pl, TINY::on, C);
  const size_t bytes = lds_total(pl->m, 3, sizeof(T), nw);
  return with_row_waves<TINY>(nw, [&](auto rwtag) -> int {
    constexpr bool RW = decltype(rwtag)::value;
    int rc;
    if ((rc = prep(k_hmc<T, TINY, RW>, bytes))) return rc;
    hipLaunchKernelGGL((k_hmc<T, TINY, RW>), dim3((unsigned)C), dim3(nw * WAVE), bytes, s, pl->m, (T*)theta, (T*)target, (T*)grad,
                       (const T*)p0, (const T*)u, (T)step, (const T*)step_vec, L, (const T*)temp, seed, iter,
                       chain_offset, (int)((flags & EY_RECOMPUTE_INITIAL_GRAD) != 0), (unsigned char*)accepted, (T*)rate,
                       (T*)hcur, (T*)hprop, run ? run->n_iters : 1, run ? (T*)run->samples : nullptr,
                       run ? (T*)run->targets : nullptr, run ? (unsigned char*)run->accepted : nullptr,
                       run ? run->accept_count : nullptr, C);
    EY_HIP(hipGetLastError());
    return (int)EY_OK;
  });
}

int ey_generic_hmc(ey_plan* pl, void* theta, void* target, void* grad, const void* p0, const void* u, double step,
                   const void* step_vec, int L, const void* temp, int64_t C, uint64_t seed, uint64_t iter,
                   uint64_t chain_offset, uint32_t flags, void* accepted, void* rate, void* hcur, void* hprop,
                   hipStream_t s, const EyRun* run) {
  return EY_TINY_DISPATCH(launch_hmc, pl, theta, target, grad, p0, u, step, step_vec, L, temp, C, seed, iter, chain_offset,
                                 flags, accepted, rate, hcur, hprop, s, run);
}

template <typename T, class TINY>
static int launch_leapfrog(ey_plan* pl, void* theta, void* p, double step, const void* step_vec, int L, const void* temp,
                           int64_t C, void* target, void* grad, hipStream_t s) {
  const int nw = row_waves_for(pl, TINY::on, C);
  const size_t bytes = lds_total(pl->m, 3, sizeof(T), nw);
  return with_row_waves<TINY>(nw, [&](auto rwtag) -> int {
    constexpr bool RW = decltype(rwtag)::value;
    int rc;
    if ((rc = prep(k_leapfrog<T, TINY, RW>, bytes))) return rc;
    hipLaunchKernelGGL((k_leapfrog<T, TINY, RW>), dim3((unsigned)C), dim3(nw * WAVE), bytes, s, pl->m, (T*)theta, (T*)p, (T)step,
                       (const T*)step_vec, L, (const T*)temp, (T*)target, (T*)grad);
    EY_HIP(hipGetLastError());
    return (int)EY_OK;
  });
}

int ey_generic_leapfrog(ey_plan* pl, void* theta, void* p, double step, const void* step_vec, int L, const void* temp,
                        int64_t C, void* target, void* grad, hipStream_t s) {
  return EY_TINY_DISPATCH(launch_leapfrog, pl, theta, p, step, step_vec, L, temp, C, target, grad, s);
}

template <typename T, class TINY>
static int launch_mala(ey_plan* pl, void* theta, void* target, void* grad, const void* z, const void* u, double step,
                       const void* step_vec, const void* temp, int64_t C, uint64_t seed, uint64_t iter,
                       uint64_t chain_offset, void* accepted, void* log_rate, hipStream_t s, const EyRun* run) {
  const EyRun one = {1, nullptr, nullptr, nullptr, nullptr};
  const int nw = row_waves_for(pl, TINY::on, C);
  const size_t bytes = lds_total(pl->m, 4, sizeof(T), nw);
  return with_row_waves<TINY>(nw, [&](auto rwtag) -> int {
    constexpr bool RW = decltype(rwtag)::value;
    int rc;
    if ((rc = prep(k_mala<T, TINY, RW>, bytes))) return rc;
    // scale = np.sqrt(step) on the python float, then cast to the model dtype (mala.py:39)
    hipLaunchKernelGGL((k_mala<T, TINY, RW>), dim3((unsigned)C), dim3(nw * WAVE), bytes, s, pl->m, (T*)theta, (T*)target, (T*)grad,
                       (const T*)z, (const T*)u, (T)step, (T)sqrt(step), (const T*)step_vec, (const T*)temp, seed, iter,
                       chain_offset, (unsigned char*)accepted, (T*)log_rate, run ? *run : one, C);
    EY_HIP(hipGetLastError());
    return (int)EY_OK;
  });
}

int ey_generic_mala(ey_plan* pl, void* theta, void* target, void* grad, const void* z, const void* u, double step,
                    const void* step_vec, const void* temp, int64_t C, uint64_t seed, uint64_t iter,
                    uint64_t chain_offset, void* accepted, void* log_rate, hipStream_t s, const EyRun* run) {
  return EY_TINY_DISPATCH(launch_mala, pl, theta, target, grad, z, u, step, step_vec, temp, C, seed, iter,
                                                  chain_offset, accepted, log_rate, s, run);
}

template <typename T, class TINY>
static int launch_mh(ey_plan* pl, void* theta, void* target, const void* z, const void* u, const void* scale,
                     const void* temp, int64_t C, uint64_t seed, uint64_t iter, uint64_t chain_offset, void* accepted,
                     void* log_rate, hipStream_t s, const EyRun* run) {
  const EyRun one = {1, nullptr, nullptr, nullptr, nullptr};
  const int nw = row_waves_for(pl, TINY::on, C);
  const size_t bytes = lds_total(pl->m, 2, sizeof(T), nw);
  return with_row_waves<TINY>(nw, [&](auto rwtag) -> int {
    constexpr bool RW = decltype(rwtag)::value;
    int rc;
    if ((rc = prep(k_mh<T, TINY, RW>, bytes))) return rc;
    hipLaunchKernelGGL((k_mh<T, TINY, RW>), dim3((unsigned)C), dim3(nw * WAVE), bytes, s, pl->m, (T*)theta, (T*)target, (const T*)z,
                       (const T*)u, (const T*)scale, (const T*)temp, seed, iter, chain_offset, (unsigned char*)accepted,
                       (T*)log_rate, run ? *run : one, C);
    EY_HIP(hipGetLastError());
    return (int)EY_OK;
  });
}

int ey_generic_mh(ey_plan* pl, void* theta, void* target, const void* z, const void* u, const void* scale,
                  const void* temp, int64_t C, uint64_t seed, uint64_t iter, uint64_t chain_offset, void* accepted,
                  void* log_rate, hipStream_t s, const EyRun* run) {
  return EY_TINY_DISPATCH(launch_mh, pl, theta, target, z, u, scale, temp, C, seed, iter, chain_offset,
                                                accepted, log_rate, s, run);
}
