// Batched chain diagnostics on the device: the initial-sequence estimator of the asymptotic
// variance (eeyore/stats/inse_mc_cov.py:9-83, the reference's O(n^2 p^2) Python double loop) for every
// (chain, parameter) series of a stored run at once, one parameter at a time (p = 1).
//
// For p = 1 the reference's matrices are scalars:
//   gam(l)  = (1/n) sum_{i < n-l} xc[i] xc[i+l]            (:24-31, xc = x - mean(x), :10)
//   Gam_m   = gam(2m) + gam(2m+1)                          (:33-34; the symmetrisation is the identity)
//   Sig     = -gam(0) + 2 Gam_0, then += 2 Gam_m           (:36-39) until Sig is positive definite, i.e. > 0 (:41-43,
//             eeyore/linalg/is_pos_def.py:3-11); never within floor(n/2) lag pairs => 'Not enough samples' (:45-46)
//   then keep adding 2 Gam_m while det(Sig) = Sig strictly increases (:50-72)
//   adjust=True adds -2 min(eig(Gam_m), 0) for the accepted m (:74-80); an accepted m has Gam_m > 0, so for p = 1 the
//   adjustment is identically zero and is not computed.
// The unbiased sample variance sum xc^2 / (n-1) (eeyore/stats/cov.py:5-15) comes out of the same pass, so that
// multi_ess' n * (det cov / det mc_cov)^(1/p) (eeyore/stats/multi_ess.py:6-14) is one division away.
//
// Layout: x [n, S] row-major, S = chains * parameters (a ChainBuffer's [iterations, C, P] as stored): threads of a wave
// read adjacent series of one iteration.  A 256-thread workgroup stages BS series (all n values, centred, in double or
// float as the input) in LDS as [i][BS]; 256/BS threads share the lag sums of a series.  Every lag pair costs 2n
// multiply-adds per series from LDS; the loop ends when every series of the workgroup has stopped.  All sums in double.
#include <algorithm>
#include <cstdlib>

#include "ey_common.h"

#define ST_THREADS 256
#define ST_LDS_BYTES (144 * 1024)

template <typename T, int BS>
__global__ void __launch_bounds__(ST_THREADS) k_inse_univariate(const T* __restrict__ x, int64_t n, int64_t S,
                                                                double* __restrict__ sig2, double* __restrict__ var,
                                                                int* __restrict__ pairs) {
  constexpr int TPS = ST_THREADS / BS;  // threads per series
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* xs = reinterpret_cast<T*>(smem_raw);  // [n][BS]
  static_assert(BS <= 64 && 64 % BS == 0, "a wave holds whole groups of BS series");
  __shared__ double red[4][ST_THREADS / 64][BS];
  const int tid = threadIdx.x;
  const int s = tid % BS, t = tid / BS;
  const int64_t s0 = (int64_t)blockIdx.x * BS;
  const bool valid = s0 + s < S;
  const int ni = (int)n;

  // combine the TPS partial sums of every series: lanes of one series are BS apart inside a wave, then across waves
  auto series_sum4 = [&](double (&v)[4]) {
#pragma unroll
    for (int o = BS; o < 64; o <<= 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] += __shfl_xor(v[q], o, 64);
    }
    const int wave = tid >> 6, lane = tid & 63;
    __syncthreads();  // the previous round's reads of `red` are done
    if (lane < BS) {
#pragma unroll
      for (int q = 0; q < 4; ++q) red[q][wave][lane] = v[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double r = 0.0;
#pragma unroll
      for (int w = 0; w < ST_THREADS / 64; ++w) r += red[q][w][s];
      v[q] = r;
    }
  };

  // ---- stage and centre (inse_mc_cov.py:10)
  double acc4[4] = {0.0, 0.0, 0.0, 0.0};
  for (int i = t; i < ni; i += TPS) {
    const T v = valid ? x[(int64_t)i * S + s0 + s] : T(0);
    xs[i * BS + s] = v;
    acc4[0] += (double)v;
  }
  series_sum4(acc4);
  const double mean = acc4[0] / (double)ni;
  for (int i = t; i < ni; i += TPS) xs[i * BS + s] = (T)((double)xs[i * BS + s] - mean);
  __syncthreads();

  const int ub = ni / 2;  // floor(n/2), :14
  double Sig = 0.0, last = 0.0, gam_zero = 0.0;
  int state = valid ? 0 : 2;  // 0: looking for the first positive Sig, 1: extending, 2: stopped
  int used = 0;
  // two lag pairs (four consecutive lags) per round: five LDS reads feed four multiply-adds
  for (int m = 0; m < ub; m += 2) {
    double g[4] = {0.0, 0.0, 0.0, 0.0};
    if (state != 2) {
      const int l0 = 2 * m;
      const int full = ni - l0 - 3;  // i < full: all four partners exist
      int i = t;
      for (; i < full; i += TPS) {
        const T* q = xs + i * BS + s;
        const double a = (double)q[0];
        g[0] += a * (double)q[l0 * BS];
        g[1] += a * (double)q[(l0 + 1) * BS];
        g[2] += a * (double)q[(l0 + 2) * BS];
        g[3] += a * (double)q[(l0 + 3) * BS];
      }
      for (; i < ni - l0; i += TPS) {  // the last three start points: partners run out one by one
        const T* q = xs + i * BS + s;
        const double a = (double)q[0];
        g[0] += a * (double)q[l0 * BS];
        if (i + l0 + 1 < ni) g[1] += a * (double)q[(l0 + 1) * BS];
        if (i + l0 + 2 < ni) g[2] += a * (double)q[(l0 + 2) * BS];
      }
    }
    series_sum4(g);
#pragma unroll
    for (int q = 0; q < 4; ++q) g[q] /= (double)ni;
    if (m == 0) gam_zero = g[0];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int mm = m + half;
      if (mm >= ub) break;
      const double Gam = g[2 * half] + g[2 * half + 1];  // :33-34
      if (state == 0) {
        Sig = (mm == 0) ? (-g[0] + 2.0 * Gam) : (Sig + 2.0 * Gam);  // :36-39
        if (Sig > 0.0) {  // positive definite (Cholesky succeeds), :41-43
          state = 1;
          last = Sig;
          used = mm + 1;
        }
      } else if (state == 1) {
        const double Sig1 = Sig + 2.0 * Gam;  // :62
        if (Sig1 <= last) {  // :64-65
          state = 2;
        } else {
          Sig = Sig1;
          last = Sig1;
          used = mm + 1;
        }
      }
    }
    if (!__syncthreads_or(state != 2)) break;  // every series of this workgroup has stopped
  }
  if (valid && t == 0) {
    const bool enough = state != 0;  // state 0 after the loop: 'Not enough samples' (:45-46)
    sig2[s0 + s] = enough ? Sig : __builtin_nan("");
    var[s0 + s] = ni > 1 ? gam_zero * (double)ni / (double)(ni - 1) : __builtin_nan("");
    if (pairs) pairs[s0 + s] = enough ? used : -1;
  }
}

template <typename T, int BS>
static int launch_inse(const void* x, int64_t n, int64_t S, double* sig2, double* var, int* pairs, hipStream_t s) {
  const size_t bytes = (size_t)n * BS * sizeof(T);
  // per launch: function attributes are per device (cheap next to the kernel)
  EY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_inse_univariate<T, BS>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, ST_LDS_BYTES));
  const unsigned grid = (unsigned)((S + BS - 1) / BS);
  hipLaunchKernelGGL((k_inse_univariate<T, BS>), dim3(grid), dim3(ST_THREADS), bytes, s, (const T*)x, n, S, sig2, var,
                     pairs);
  EY_HIP(hipGetLastError());
  return EY_OK;
}

template <typename T>
static int dispatch_inse(const void* x, int64_t n, int64_t S, double* sig2, double* var, int* pairs, hipStream_t s) {
  const size_t per_series = (size_t)n * sizeof(T);
  if (16 * per_series <= ST_LDS_BYTES) return launch_inse<T, 16>(x, n, S, sig2, var, pairs, s);
  if (4 * per_series <= ST_LDS_BYTES) return launch_inse<T, 4>(x, n, S, sig2, var, pairs, s);
  if (per_series <= ST_LDS_BYTES) return launch_inse<T, 1>(x, n, S, sig2, var, pairs, s);
  EY_FAIL(EY_ERR_UNSUPPORTED, "ey_inse_univariate: a series of this length does not fit LDS (n <= 36864 for f32, 18432 "
                              "for f64)");
}

extern "C" int ey_inse_univariate(const void* x, int64_t n, int64_t S, int dtype, void* sig2, void* var, void* num_pairs,
                                  void* stream) {
  if (!x || !sig2 || !var) EY_FAIL(EY_ERR_INVALID, "ey_inse_univariate: null argument");
  if (dtype != EY_F32 && dtype != EY_F64) EY_FAIL(EY_ERR_INVALID, "ey_inse_univariate: bad dtype");
  if (n < 2) EY_FAIL(EY_ERR_INVALID, "ey_inse_univariate: at least two iterations are needed");
  if (n > 0x7fffffff / 2) EY_FAIL(EY_ERR_INVALID, "ey_inse_univariate: too many iterations");
  if (S <= 0) return EY_OK;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == EY_F32) return dispatch_inse<float>(x, n, S, (double*)sig2, (double*)var, (int*)num_pairs, s);
  return dispatch_inse<double>(x, n, S, (double*)sig2, (double*)var, (int*)num_pairs, s);
}

// ----------------------------------------------------------------------------------------------- multivariate (p > 1)
// The reference's estimator as it stands (eeyore/stats/inse_mc_cov.py:9-83, adjust = False) for C chains at once, one
// 256-thread workgroup per chain: the chain's n x p centred samples are staged in LDS; for every lag pair m the
// workgroup forms gam(2m) and gam(2m+1) together (thread (a, b, slice) sums its slice of i for the p^2 entries of both
// matrices, wave shuffles + one LDS exchange combine the slices), then one thread runs the reference's p x p logic:
// symmetrise (:33-34), accumulate (:36-39, :62), positive definiteness as a Cholesky attempt on an exactly symmetric
// matrix (:41, eeyore/linalg/is_pos_def.py:3-11), determinant by elimination with partial pivoting (:48, :63).
// Also returns the unbiased sample covariance (eeyore/stats/cov.py:5-15), so that multi_ess (eeyore/stats/multi_ess.py:
// 6-14) and the within-chain part W of multi_rhat (eeyore/stats/multi_rhat.py:14-22) need nothing else.
// Layout: x [n, C, p] (a chain buffer as stored) with element strides (sn, sc); sums in double whatever the input.
#define MV_PMAX 16

__device__ inline bool mv_chol_ok(const double* a, int p) {  // is_pos_def: symmetric (exactly) and Cholesky succeeds
  double l[MV_PMAX * MV_PMAX];
  for (int i = 0; i < p; ++i)
    for (int j = 0; j < i; ++j)
      if (a[i * p + j] != a[j * p + i]) return false;
  for (int j = 0; j < p; ++j) {
    double d = a[j * p + j];
    for (int k = 0; k < j; ++k) d -= l[j * p + k] * l[j * p + k];
    if (!(d > 0.0)) return false;
    const double dj = sqrt(d);
    l[j * p + j] = dj;
    for (int i = j + 1; i < p; ++i) {
      double v = a[i * p + j];
      for (int k = 0; k < j; ++k) v -= l[i * p + k] * l[j * p + k];
      l[i * p + j] = v / dj;
    }
  }
  return true;
}
__device__ inline double mv_det(const double* a, int p) {  // LU with partial pivoting, as torch.det
  double m[MV_PMAX * MV_PMAX];
  for (int i = 0; i < p * p; ++i) m[i] = a[i];
  double det = 1.0;
  for (int k = 0; k < p; ++k) {
    int piv = k;
    double best = fabs(m[k * p + k]);
    for (int i = k + 1; i < p; ++i)
      if (fabs(m[i * p + k]) > best) { best = fabs(m[i * p + k]); piv = i; }
    if (best == 0.0) return 0.0;
    if (piv != k) {
      for (int j = 0; j < p; ++j) { const double t = m[k * p + j]; m[k * p + j] = m[piv * p + j]; m[piv * p + j] = t; }
      det = -det;
    }
    det *= m[k * p + k];
    for (int i = k + 1; i < p; ++i) {
      const double f = m[i * p + k] / m[k * p + k];
      for (int j = k + 1; j < p; ++j) m[i * p + j] -= f * m[k * p + j];
    }
  }
  return det;
}

template <typename T>
__global__ void __launch_bounds__(ST_THREADS) k_inse_multivariate(const T* __restrict__ x, int64_t n, int64_t C, int p,
                                                                  int64_t sn, int64_t sc, double* __restrict__ sig,
                                                                  double* __restrict__ cov, double* __restrict__ mean_o,
                                                                  int* __restrict__ pairs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* xs = reinterpret_cast<double*>(smem_raw);  // [n][p] centred
  __shared__ double gsh[2][MV_PMAX * MV_PMAX];
  __shared__ double mean_s[MV_PMAX];
  __shared__ int go_on;
  const int tid = threadIdx.x;
  const int64_t c = blockIdx.x;
  const int ni = (int)n, pp = p * p;
  const T* xc = x + c * sc;
  // ---- stage, mean (inse_mc_cov.py:10), centre
  for (int i = tid; i < ni * p; i += ST_THREADS) xs[i] = (double)xc[(int64_t)(i / p) * sn + (i % p)];
  __syncthreads();
  if (tid < p) {
    double s = 0.0;
    for (int i = 0; i < ni; ++i) s += xs[i * p + tid];
    mean_s[tid] = s / (double)ni;
    if (mean_o) mean_o[c * p + tid] = mean_s[tid];
  }
  __syncthreads();
  for (int i = tid; i < ni * p; i += ST_THREADS) xs[i] -= mean_s[i % p];
  __syncthreads();
  // thread -> (entry e = a p + b, slice): slices of i are spread over the threads that share an entry
  const int nsl = ST_THREADS / pp > 0 ? ST_THREADS / pp : 1;   // slices per entry (pp <= 256)
  const int e = tid % pp, sl = tid / pp;
  const bool worker = sl < nsl;
  const int a = e / p, b = e % p;
  // gam(l0) and gam(l0 + 1) of every entry into gsh[0..1]
  auto lag_pair = [&](int l0) {
    double g0 = 0.0, g1 = 0.0;
    if (worker) {
      for (int i = sl; i < ni - l0; i += nsl) {
        const double xa = xs[i * p + a];
        g0 += xa * xs[(i + l0) * p + b];
        if (i + l0 + 1 < ni) g1 += xa * xs[(i + l0 + 1) * p + b];
      }
    }
    // combine the slices in a fixed order (reproducible): every worker leaves its partial, the entry's first thread adds
    __syncthreads();
    __shared__ double part[2][ST_THREADS];
    part[0][tid] = worker ? g0 : 0.0;
    part[1][tid] = worker ? g1 : 0.0;
    __syncthreads();
    if (tid < pp) {
      double s0 = 0.0, s1 = 0.0;
      for (int k = 0; k < nsl; ++k) { s0 += part[0][k * pp + tid]; s1 += part[1][k * pp + tid]; }
      gsh[0][tid] = s0 / (double)ni;
      gsh[1][tid] = s1 / (double)ni;
    }
    __syncthreads();
  };

  const int ub = ni / 2;
  // thread 0's state of the reference's loops
  double Sig[MV_PMAX * MV_PMAX], Cand[MV_PMAX * MV_PMAX];
  double last = 0.0;
  int state = 0, used = 0;  // 0: looking for the first positive definite Sig, 1: extending, 2: stopped
  for (int m = 0; m < ub; ++m) {
    lag_pair(2 * m);
    if (tid == 0) {
      if (m == 0 && cov) {  // unbiased sample covariance: gam(0) n / (n - 1)
        for (int k = 0; k < pp; ++k) cov[c * pp + k] = gsh[0][k] * (double)ni / (double)(ni - 1);
      }
      for (int i = 0; i < p; ++i)
        for (int j = 0; j < p; ++j) {
          const double Gij = gsh[0][i * p + j] + gsh[1][i * p + j], Gji = gsh[0][j * p + i] + gsh[1][j * p + i];
          const double G = (Gij + Gji) / 2.0;                                                   // :33-34
          Cand[i * p + j] = (m == 0) ? (-gsh[0][i * p + j] + 2.0 * G) : (Sig[i * p + j] + 2.0 * G);  // :36-39 / :62
        }
      if (state == 0) {
        for (int k = 0; k < pp; ++k) Sig[k] = Cand[k];
        if (mv_chol_ok(Sig, p)) {  // :41-43
          state = 1;
          last = mv_det(Sig, p);   // :48
          used = m + 1;
        }
      } else {
        const double dtm = mv_det(Cand, p);  // :63
        if (dtm <= last) {                   // :64-65
          state = 2;
        } else {
          for (int k = 0; k < pp; ++k) Sig[k] = Cand[k];
          last = dtm;
          used = m + 1;
        }
      }
      go_on = state != 2;
    }
    __syncthreads();
    if (!go_on) break;
  }
  if (tid == 0) {
    const bool enough = state != 0;  // 'Not enough samples' (:45-46)
    for (int k = 0; k < pp; ++k) sig[c * pp + k] = enough ? Sig[k] : __builtin_nan("");
    if (pairs) pairs[c] = enough ? used : -1;
  }
}

// ---- the same estimator for wider chains, 16 < p <= 64 (and any n): config 2's MLP(2-3-2-1) already has 20 parameters.
// One 256-thread workgroup per chain.  The centred samples do not fit LDS (n x p doubles), so the workgroup leaves them in
// a global workspace (xc [C][n][64] doubles, columns beyond p zero) and streams them back through LDS in tiles of MW_TR
// rows: thread (ta, tb) owns the 4 x 4 block of entries a = 4 ta .., b = 4 tb .. of BOTH lag matrices of a pair and sums
// each entry over i in ascending order -- the order in which the reference adds its outer products (:24-31) -- so the sums
// do not depend on the launch.  The p x p logic then runs on the whole workgroup instead of one thread: symmetrise and
// accumulate through LDS, positive definiteness as a right-looking Cholesky attempt on an exactly symmetric matrix
// (two barriers per column), the determinant by LU with partial pivoting (the first largest pivot, as LAPACK's getrf under
// torch.det takes it; three barriers per column).  Matrices lie in LDS with a row stride of 65 doubles (columns are read
// conflict-free).
#define MW_P 64
#define MW_LD 65
#define MW_TR 16
__device__ inline bool mw_chol_ok(double* w, int p, int tid, int* flag) {  // w is destroyed; uniform result
  if (tid == 0) *flag = 1;
  __syncthreads();
  bool sym = true;
  for (int e = tid; e < p * p; e += ST_THREADS) {
    const int i = e / p, j = e - i * p;
    if (w[i * MW_LD + j] != w[j * MW_LD + i]) sym = false;
  }
  if (!sym) *flag = 0;
  __syncthreads();
  if (!*flag) return false;
  for (int j = 0; j < p; ++j) {
    const double d = w[j * MW_LD + j];
    if (!(d > 0.0)) return false;  // every thread reads the same value: uniform
    const double dj = sqrt(d);
    __syncthreads();               // everyone has read the diagonal element before the column is scaled
    for (int i = j + 1 + tid; i < p; i += ST_THREADS) w[i * MW_LD + j] /= dj;
    __syncthreads();
    const int r = p - 1 - j;       // trailing square, lower triangle
    for (int e = tid; e < r * r; e += ST_THREADS) {
      const int i = j + 1 + e / r, k = j + 1 + e % r;
      if (k <= i) w[i * MW_LD + k] -= w[i * MW_LD + j] * w[k * MW_LD + j];
    }
    __syncthreads();
  }
  return true;
}
__device__ inline double mw_det(double* w, int p, int tid, int* piv_s) {  // w is destroyed; uniform result
  double det = 1.0;
  for (int k = 0; k < p; ++k) {
    if (tid < 64) {  // wave 0: lane i looks at row i; the first largest |w[i][k]|, i >= k
      double v = (tid >= k && tid < p) ? fabs(w[tid * MW_LD + k]) : -1.0;
      int idx = tid;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const double v2 = __shfl_xor(v, o, 64);
        const int i2 = __shfl_xor(idx, o, 64);
        if (v2 > v || (v2 == v && i2 < idx)) { v = v2; idx = i2; }
      }
      if (tid == 0) *piv_s = idx;
    }
    __syncthreads();
    const int piv = *piv_s;
    if (w[piv * MW_LD + k] == 0.0) return 0.0;  // uniform
    if (piv != k) {
      __syncthreads();  // everyone has read the pivot element
      for (int j = tid; j < p; j += ST_THREADS) {
        const double t = w[k * MW_LD + j];
        w[k * MW_LD + j] = w[piv * MW_LD + j];
        w[piv * MW_LD + j] = t;
      }
      det = -det;
      __syncthreads();
    }
    const double pv = w[k * MW_LD + k];
    det *= pv;
    const int r = p - 1 - k;
    for (int e = tid; e < r * r; e += ST_THREADS) {
      const int i = k + 1 + e / r, j = k + 1 + e % r;
      const double f = w[i * MW_LD + k] / pv;
      w[i * MW_LD + j] -= f * w[k * MW_LD + j];
    }
    __syncthreads();
  }
  return det;
}

template <typename T>
__global__ void __launch_bounds__(ST_THREADS) k_inse_mv_wide(const T* __restrict__ x, int64_t n, int64_t C, int p, int64_t sn,
                                                             int64_t sc, double* __restrict__ xcw, double* __restrict__ sig,
                                                             double* __restrict__ cov, double* __restrict__ mean_o,
                                                             int* __restrict__ pairs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* Sig = reinterpret_cast<double*>(smem_raw);  // [64][65] each
  double* Cand = Sig + MW_P * MW_LD;
  double* Wk = Cand + MW_P * MW_LD;
  double* sa = Wk + MW_P * MW_LD;                      // [MW_TR][64]     rows i0 ..
  double* sb = sa + MW_TR * MW_P;                      // [MW_TR + 1][64] rows i0 + l0 ..
  __shared__ double mean_s[MW_P];
  __shared__ double part[4][MW_P];
  __shared__ int flag_s, piv_s;
  const int tid = threadIdx.x;
  const int64_t c = blockIdx.x;
  const int ni = (int)n;
  const T* xc = x + c * sc;
  double* xw = xcw + c * (int64_t)ni * MW_P;
  // ---- mean (inse_mc_cov.py:10): column j by four threads over interleaved rows, combined in a fixed order; centre
  {
    const int j = tid & 63, q = tid >> 6;
    double s = 0.0;
    if (j < p)
      for (int i = q; i < ni; i += 4) s += (double)xc[(int64_t)i * sn + j];
    part[q][j] = s;
    __syncthreads();
    if (tid < MW_P) {
      const double m = tid < p ? (((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid]) / (double)ni : 0.0;
      mean_s[tid] = m;
      if (mean_o && tid < p) mean_o[c * p + tid] = m;
    }
    __syncthreads();
    for (int e = tid; e < ni * MW_P; e += ST_THREADS) {
      const int i = e >> 6, jj = e & 63;
      xw[e] = jj < p ? (double)xc[(int64_t)i * sn + jj] - mean_s[jj] : 0.0;
    }
    __syncthreads();  // (this workgroup's own global writes are read back below: same CU, in order)
  }
  const int ta = tid >> 4, tb = tid & 15;
  double g0[4][4], g1[4][4];
  // gam(l0) and gam(l0 + 1), the entries of this thread, summed over i ascending
  auto lag_pair = [&](int l0) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v) { g0[u][v] = 0.0; g1[u][v] = 0.0; }
    const int cnt = ni - l0;  // rows i = 0 .. cnt - 1 enter gam(l0); gam(l0 + 1) ends one earlier (its partner row is zero)
    for (int i0 = 0; i0 < cnt; i0 += MW_TR) {
      __syncthreads();
      for (int e = tid; e < MW_TR * MW_P; e += ST_THREADS) {
        const int i = i0 + (e >> 6);
        sa[e] = i < cnt ? xw[(int64_t)i * MW_P + (e & 63)] : 0.0;
      }
      for (int e = tid; e < (MW_TR + 1) * MW_P; e += ST_THREADS) {
        const int i = i0 + l0 + (e >> 6);
        sb[e] = i < ni ? xw[(int64_t)i * MW_P + (e & 63)] : 0.0;
      }
      __syncthreads();
#pragma unroll 4
      for (int ii = 0; ii < MW_TR; ++ii) {
        double xa[4], xb0[4], xb1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          xa[u] = sa[ii * MW_P + 4 * ta + u];
          xb0[u] = sb[ii * MW_P + 4 * tb + u];
          xb1[u] = sb[(ii + 1) * MW_P + 4 * tb + u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            g0[u][v] += xa[u] * xb0[v];
            g1[u][v] += xa[u] * xb1[v];
          }
      }
    }
  };
  const int ub = ni / 2;
  double last = 0.0;
  int state = 0, used = 0;  // 0: looking for the first positive definite Sig, 1: extending, 2: stopped (uniform)
  for (int m = 0; m < ub && state != 2; ++m) {
    lag_pair(2 * m);
    // gam0 -> Cand, gam1 -> Wk (divided by n, :28, :31), then the candidate of every entry from both triangles
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int a = 4 * ta + u, b = 4 * tb + v;
        Cand[a * MW_LD + b] = g0[u][v] / (double)ni;
        Wk[a * MW_LD + b] = g1[u][v] / (double)ni;
      }
    __syncthreads();
    double cnd[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int a = 4 * ta + u, b = 4 * tb + v;
        const double G0ab = Cand[a * MW_LD + b];
        const double Gab = G0ab + Wk[a * MW_LD + b], Gba = Cand[b * MW_LD + a] + Wk[b * MW_LD + a];
        const double G = (Gab + Gba) / 2.0;                                              // :33-34
        cnd[u][v] = (m == 0) ? (-G0ab + 2.0 * G) : (Sig[a * MW_LD + b] + 2.0 * G);       // :36-39 / :62
        if (m == 0 && cov && a < p && b < p) cov[c * p * p + a * p + b] = G0ab * (double)ni / (double)(ni - 1);
      }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int a = 4 * ta + u, b = 4 * tb + v;
        Cand[a * MW_LD + b] = cnd[u][v];
        Wk[a * MW_LD + b] = cnd[u][v];
        if (state == 0) Sig[a * MW_LD + b] = cnd[u][v];
      }
    __syncthreads();
    if (state == 0) {
      if (mw_chol_ok(Wk, p, tid, &flag_s)) {  // :41-43
        __syncthreads();
        for (int e = tid; e < MW_P * MW_LD; e += ST_THREADS) Wk[e] = Sig[e];
        __syncthreads();
        last = mw_det(Wk, p, tid, &piv_s);    // :48
        state = 1;
        used = m + 1;
      }
    } else {
      const double dtm = mw_det(Wk, p, tid, &piv_s);  // :63
      if (dtm <= last) {                              // :64-65
        state = 2;
      } else {
        __syncthreads();
        for (int e = tid; e < MW_P * MW_LD; e += ST_THREADS) Sig[e] = Cand[e];
        last = dtm;
        used = m + 1;
      }
    }
    __syncthreads();
  }
  const bool enough = state != 0;  // 'Not enough samples' (:45-46)
  for (int e = tid; e < p * p; e += ST_THREADS) {
    const int a = e / p, b = e - a * p;
    sig[c * p * p + e] = enough ? Sig[a * MW_LD + b] : __builtin_nan("");
  }
  if (pairs && tid == 0) pairs[c] = enough ? used : -1;
}
#define MW_LDS_BYTES ((3 * MW_P * MW_LD + (2 * MW_TR + 1) * MW_P) * sizeof(double))

extern "C" int ey_inse_multivariate(const void* x, int64_t n, int64_t C, int64_t p, int64_t stride_n, int64_t stride_c,
                                    int dtype, void* sig, void* cov, void* mean, void* num_pairs, void* stream) {
  if (!x || !sig) EY_FAIL(EY_ERR_INVALID, "ey_inse_multivariate: null argument");
  if (dtype != EY_F32 && dtype != EY_F64) EY_FAIL(EY_ERR_INVALID, "ey_inse_multivariate: bad dtype");
  if (n < 2) EY_FAIL(EY_ERR_INVALID, "ey_inse_multivariate: at least two iterations are needed");
  if (p < 1 || p > MW_P) EY_FAIL(EY_ERR_UNSUPPORTED, "ey_inse_multivariate: 1 <= p <= 64 (use ey_inse_univariate per parameter beyond)");
  if (n > (0x7fffffff - 256) / MW_P) EY_FAIL(EY_ERR_INVALID, "ey_inse_multivariate: too many iterations");  // (32-bit loop indices up to n * 64 + the block size)
  const size_t bytes = (size_t)n * (size_t)p * sizeof(double);
  if (p > MV_PMAX || bytes > ST_LDS_BYTES) {  // the wide form: the centred chains in a workspace, matrices in LDS
    if (C <= 0) return EY_OK;
    hipStream_t s = (hipStream_t)stream;
    // (everything that can fail before the workspace exists comes first; nothing returns between its allocation and its release)
    EY_HIP(hipFuncSetAttribute(dtype == EY_F32 ? reinterpret_cast<const void*>(k_inse_mv_wide<float>)
                                               : reinterpret_cast<const void*>(k_inse_mv_wide<double>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)MW_LDS_BYTES));
    // The centred chains of one launch lie in a workspace of 64 columns per iteration whatever p is: bounded (EY_MV_WORKSPACE_MB,
    // 1 GiB by default), the chains going through it in as many launches as that takes (4096 chains x 10 000 iterations would
    // otherwise ask HIP's pool -- not torch's cache, which holds the device -- for 21 GB at once).
    static const size_t cap = [] { const char* e = getenv("EY_MV_WORKSPACE_MB"); return (size_t)(e && atoi(e) > 0 ? atoi(e) : 1024) << 20; }();
    const size_t per_chain = (size_t)n * MW_P * sizeof(double);
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(C, (int64_t)(cap / per_chain)));
    double* xcw = nullptr;
    EY_HIP(hipMallocAsync((void**)&xcw, (size_t)chunk * per_chain, s));
    const size_t es = dtype == EY_F32 ? sizeof(float) : sizeof(double);
    hipError_t le = hipSuccess;
    for (int64_t c0 = 0; c0 < C && le == hipSuccess; c0 += chunk) {
      const int64_t cn = std::min<int64_t>(chunk, C - c0);
      const char* xc = (const char*)x + (size_t)c0 * (size_t)stride_c * es;
      double* sg = (double*)sig + c0 * p * p;
      double* cv = cov ? (double*)cov + c0 * p * p : nullptr;
      double* mn = mean ? (double*)mean + c0 * p : nullptr;
      int* np_ = num_pairs ? (int*)num_pairs + c0 : nullptr;
      if (dtype == EY_F32)
        hipLaunchKernelGGL(k_inse_mv_wide<float>, dim3((unsigned)cn), dim3(ST_THREADS), MW_LDS_BYTES, s, (const float*)xc, n, cn,
                           (int)p, stride_n, stride_c, xcw, sg, cv, mn, np_);
      else
        hipLaunchKernelGGL(k_inse_mv_wide<double>, dim3((unsigned)cn), dim3(ST_THREADS), MW_LDS_BYTES, s, (const double*)xc, n, cn,
                           (int)p, stride_n, stride_c, xcw, sg, cv, mn, np_);
      le = hipGetLastError();
    }
    (void)hipFreeAsync(xcw, s);
    EY_HIP(le);
    return EY_OK;
  }
  if (C <= 0) return EY_OK;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == EY_F32) {
    EY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_inse_multivariate<float>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, ST_LDS_BYTES));
    hipLaunchKernelGGL(k_inse_multivariate<float>, dim3((unsigned)C), dim3(ST_THREADS), bytes, s, (const float*)x, n, C,
                       (int)p, stride_n, stride_c, (double*)sig, (double*)cov, (double*)mean, (int*)num_pairs);
  } else {
    EY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_inse_multivariate<double>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, ST_LDS_BYTES));
    hipLaunchKernelGGL(k_inse_multivariate<double>, dim3((unsigned)C), dim3(ST_THREADS), bytes, s, (const double*)x, n, C,
                       (int)p, stride_n, stride_c, (double*)sig, (double*)cov, (double*)mean, (int*)num_pairs);
  }
  EY_HIP(hipGetLastError());
  return EY_OK;
}
