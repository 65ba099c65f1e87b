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

extern "C" int ey_inse_multivariate(const void* x, int64_t n, int64_t C, int64_t p, int64_t stride_n, int64_t stride_c,
                                    int dtype, void* sig, void* cov, void* mean, void* num_pairs, void* stream) {
  if (!x || !sig) EY_FAIL(EY_ERR_INVALID, "ey_inse_multivariate: null argument");
  if (dtype != EY_F32 && dtype != EY_F64) EY_FAIL(EY_ERR_INVALID, "ey_inse_multivariate: bad dtype");
  if (n < 2) EY_FAIL(EY_ERR_INVALID, "ey_inse_multivariate: at least two iterations are needed");
  if (p < 1 || p > MV_PMAX) EY_FAIL(EY_ERR_UNSUPPORTED, "ey_inse_multivariate: 1 <= p <= 16 (use ey_inse_univariate per parameter beyond)");
  const size_t bytes = (size_t)n * (size_t)p * sizeof(double);
  if (bytes > ST_LDS_BYTES) EY_FAIL(EY_ERR_UNSUPPORTED, "ey_inse_multivariate: a chain of n x p doubles must fit 144 KiB of LDS");
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
