// The operand split of the bf16x3 form (x = hi + mid + lo, three bf16 pieces): is a remainder x - bf16(x) cheaper as one
// v_dot2c_f32_bf16 (packed pair (hi_a, hi_b) . (-1, 0) + a) than as an unpack (shift / mask) and a subtract, and is it the
// same number?   (run on the GPU box)
//   hipcc -O3 --offload-arch=gfx950 tools/split_probe.hip -o tools/split_probe && tools/split_probe
// (a) issue cost of v_dot2c_f32_bf16 and v_perm_b32 beside v_fma_f32, one and two waves per SIMD;
// (b) the two forms of the split on 2^26 floats of every magnitude (normal, tiny, huge, signed zero): pieces compared bit
//     for bit, and hi + mid + lo == x checked;
// (c) time of the two forms on a 16-element tile per lane, two waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct Pieces { u32x4 hi[2], mid[2], lo[2]; };

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float dot2c(unsigned a, unsigned b, float c) {
  asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(c) : "v"(a), "v"(b));
  return c;
}
#define NEG1_LO 0x0000bf80u  // (-1, 0): the low element of the pair
#define NEG1_HI 0xbf800000u  // (0, -1): the high element

template <int FORM>
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hh, unsigned& mm, unsigned& ll) {
  hh = pk_bf16(a, b);
  if (FORM == 0) {
    const float ra = a - __builtin_bit_cast(float, hh << 16), rb = b - __builtin_bit_cast(float, hh & 0xffff0000u);
    mm = pk_bf16(ra, rb);
    const float sa = ra - __builtin_bit_cast(float, mm << 16), sb = rb - __builtin_bit_cast(float, mm & 0xffff0000u);
    ll = pk_bf16(sa, sb);
  } else {
    const float ra = dot2c(hh, NEG1_LO, a), rb = dot2c(hh, NEG1_HI, b);
    mm = pk_bf16(ra, rb);
    const float sa = dot2c(mm, NEG1_LO, ra), sb = dot2c(mm, NEG1_HI, rb);
    if (FORM == 1) ll = pk_bf16(sa, sb);
    else ll = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, sb), __builtin_bit_cast(unsigned, sa), 0x07060302u);
  }
}
template <int FORM>
__device__ __forceinline__ void split16(const f32x16& v, Pieces& P) {
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      unsigned hh, mm, ll;
      split_pair<FORM>(v[8 * s + 2 * d], v[8 * s + 2 * d + 1], hh, mm, ll);
      P.hi[s][d] = hh; P.mid[s][d] = mm; P.lo[s][d] = ll;
    }
}

enum { OP_FMA = 0, OP_DOT = 1, OP_PERM = 2, OP_CVT = 3 };
template <int OP, int W>
__global__ void __launch_bounds__(256 * W, W) k_valu(int iters, float* out) {
  float v[8];
  for (int r = 0; r < 8; ++r) v[r] = threadIdx.x * 1e-3f + r;
  unsigned q = threadIdx.x * 0x01010101u;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        if (OP == OP_FMA) v[r] = __builtin_fmaf(v[r], 1.0001f, 0.5f);
        if (OP == OP_DOT) v[r] = dot2c(q, NEG1_LO, v[r]);
        if (OP == OP_PERM) {
          unsigned o;
          asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(o) : "v"(v[r]), "v"(v[(r + 1) & 7]), "v"(0x07060302u));
          v[r] = __builtin_bit_cast(float, o);
        }
        if (OP == OP_CVT) {
          unsigned o;
          asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(o) : "v"(v[r]), "v"(v[(r + 1) & 7]));
          v[r] = __builtin_bit_cast(float, o);
        }
      }
  }
  float s = 0;
  for (int r = 0; r < 8; ++r) s += v[r];
  out[blockIdx.x * 256 * W + threadIdx.x] = s;
}

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// every float bit pattern class: the exponent is drawn from the whole range (so denormals, tiny and huge values occur)
__global__ void k_check(uint64_t n, unsigned long long* bad) {
  unsigned long long diff_piece = 0, not_exact = 0, diff_lo_form = 0;
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t ua = mix((uint32_t)i * 2u + 1u), ub = mix((uint32_t)i * 2u + 2u);
    const float a = __builtin_bit_cast(float, ua), b = __builtin_bit_cast(float, ub);
    if (!(__builtin_fabsf(a) < 3e38f) || !(__builtin_fabsf(b) < 3e38f)) continue;  // inf / nan / values that round to inf
    unsigned h0, m0, l0, h1, m1, l1, h2, m2, l2;
    split_pair<0>(a, b, h0, m0, l0);
    split_pair<1>(a, b, h1, m1, l1);
    split_pair<2>(a, b, h2, m2, l2);
    if (h0 != h1 || m0 != m1 || l0 != l1) ++diff_piece;
    if (l2 != l1) ++diff_lo_form;
    const float sa = (__builtin_bit_cast(float, h2 << 16) + __builtin_bit_cast(float, m2 << 16)) + __builtin_bit_cast(float, l2 << 16);
    const float sb = (__builtin_bit_cast(float, h2 & 0xffff0000u) + __builtin_bit_cast(float, m2 & 0xffff0000u)) +
                     __builtin_bit_cast(float, l2 & 0xffff0000u);
    // (denormal inputs: the library is built with denormals on; a sum that differs is counted)
    if (sa != a || sb != b) ++not_exact;
  }
  if (diff_piece) atomicAdd(&bad[0], diff_piece);
  if (not_exact) atomicAdd(&bad[1], not_exact);
  if (diff_lo_form) atomicAdd(&bad[2], diff_lo_form);
}

template <int FORM, int W>
__global__ void __launch_bounds__(256 * W, W) k_time(int iters, float* out) {
  f32x16 v;
  for (int r = 0; r < 16; ++r) v[r] = threadIdx.x * 1e-3f + r * 0.37f;
  unsigned acc = 0;
  for (int i = 0; i < iters; ++i) {
    Pieces P;
    split16<FORM>(v, P);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int d = 0; d < 4; ++d) acc ^= P.hi[s][d] + P.mid[s][d] + P.lo[s][d];
    asm volatile("" : "+v"(acc));
#pragma unroll
    for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(v[r]));
  }
  out[blockIdx.x * 256 * W + threadIdx.x] = __builtin_bit_cast(float, acc);
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

int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * 4 * 4);
  const int iters = 2000;
  const float f1 = timeit([&] { k_valu<OP_FMA, 1><<<256, 256>>>(iters, out); });
  const double ghz_cyc = 2.22 / (f1 * 1e-3 / (iters * 128.0));  // cycles per second implied by 2.22 cycles per v_fma
  auto cyc = [&](float ms, double n) { return ms * 1e-3 * ghz_cyc / n; };
  printf("(a) cycles per instruction of one wave's stream (2 waves: until both streams are done, per instruction of one)\n");
#define ROW(NAME, OP) \
  printf("  %-18s 1 wave/SIMD %5.2f   2 waves/SIMD %5.2f\n", NAME, \
         cyc(timeit([&] { k_valu<OP, 1><<<256, 256>>>(iters, out); }), iters * 128.0), \
         cyc(timeit([&] { k_valu<OP, 2><<<256, 512>>>(iters, out); }), iters * 128.0))
  ROW("v_fma_f32", OP_FMA);
  ROW("v_dot2c_f32_bf16", OP_DOT);
  ROW("v_perm_b32", OP_PERM);
  ROW("v_cvt_pk_bf16_f32", OP_CVT);
  unsigned long long* bad;
  hipMalloc(&bad, 24);
  hipMemset(bad, 0, 24);
  const uint64_t n = 1ull << 26;
  k_check<<<1024, 256>>>(n, bad);
  unsigned long long hb[3];
  hipMemcpy(hb, bad, 24, hipMemcpyDeviceToHost);
  printf("(b) %llu pairs: pieces that differ between the subtract and the dot2c form %llu; pairs with hi + mid + lo != x %llu;\n"
         "    low pieces that differ between v_cvt_pk and v_perm packing %llu\n", (unsigned long long)n, hb[0], hb[1], hb[2]);
  printf("(c) cycles per split of a 16-element tile (per wave; 2 waves per SIMD)\n");
  printf("  unpack + subtract, three v_cvt_pk      %6.1f\n", cyc(timeit([&] { k_time<0, 2><<<256, 512>>>(iters, out); }), iters));
  printf("  v_dot2c remainders, three v_cvt_pk     %6.1f\n", cyc(timeit([&] { k_time<1, 2><<<256, 512>>>(iters, out); }), iters));
  printf("  v_dot2c remainders, low piece v_perm   %6.1f\n", cyc(timeit([&] { k_time<2, 2><<<256, 512>>>(iters, out); }), iters));
  return 0;
}
