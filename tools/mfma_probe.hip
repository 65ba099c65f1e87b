// Probe of the f32 MFMA lane/register maps on gfx950 (run on the GPU box).  Prints, for every D element,
// which A lane and which B lane produced it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k4(const float* a, const float* b, float* d) {
  int l = threadIdx.x;
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}
__global__ void k32(const float* a, const float* b, float* d) {
  int l = threadIdx.x;
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0;
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[l], b[l], c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) d[l * 16 + r] = c[r];
}
__global__ void k16(const float* a, const float* b, float* d) {
  int l = threadIdx.x;
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[l], b[l], c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}

typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ void k16d(const double* a, const double* b, double* d) {
  int l = threadIdx.x;
  f64x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[l], b[l], c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}

static void probe_f64() {
  double *a, *b, *d;
  hipMalloc(&a, 512); hipMalloc(&b, 512); hipMalloc(&d, 64 * 4 * 8);
  std::vector<double> ha(64), hb(64), hd(256), hd2(256);
  for (int g = 0; g < 4; ++g) {
    for (int l = 0; l < 64; ++l) { ha[l] = ((l >> 4) == g) ? (l + 1) : 0; hb[l] = ((l >> 4) == g) ? 1 : 0; }
    hipMemcpy(a, ha.data(), 512, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 512, hipMemcpyHostToDevice);
    k16d<<<1, 64>>>(a, b, d); hipMemcpy(hd.data(), d, 2048, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) { ha[l] = ((l >> 4) == g) ? 1 : 0; hb[l] = ((l >> 4) == g) ? (l + 1) : 0; }
    hipMemcpy(a, ha.data(), 512, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 512, hipMemcpyHostToDevice);
    k16d<<<1, 64>>>(a, b, d); hipMemcpy(hd2.data(), d, 2048, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 4; ++r) {
        int la = (int)hd[l * 4 + r] - 1, lb = (int)hd2[l * 4 + r] - 1;
        int row = 4 * (l >> 4) + r;
        if (la != row + 16 * g || lb != (l & 15) + 16 * g) { if (bad < 4) printf("  f64 16x16x4 MISMATCH lane %d reg %d: A %d B %d\n", l, r, la, lb); ++bad; }
      }
    printf("f64 16x16x4 k-slot %d mismatches vs (row=4*(lane>>4)+reg, col=lane&15): %d\n", g, bad);
  }
}

int main() {
  probe_f64();
  float *a, *b, *d;
  hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 64 * 16 * 4);
  std::vector<float> ha(64), hb(64), hd(64 * 16), hd2(64 * 16);
  // ---- 4x4x1 (16 blocks, K=1): each D element = A[la] * B[lb]
  for (int l = 0; l < 64; ++l) { ha[l] = l + 1; hb[l] = 1; }
  hipMemcpy(a, ha.data(), 256, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice);
  k4<<<1, 64>>>(a, b, d); hipMemcpy(hd.data(), d, 64 * 4 * 4, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) { ha[l] = 1; hb[l] = l + 1; }
  hipMemcpy(a, ha.data(), 256, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice);
  k4<<<1, 64>>>(a, b, d); hipMemcpy(hd2.data(), d, 64 * 4 * 4, hipMemcpyDeviceToHost);
  printf("4x4x1: D[lane][reg] <- (A lane, B lane)\n");
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int r = 0; r < 4; ++r) printf(" (%2d,%2d)", (int)hd[l * 4 + r] - 1, (int)hd2[l * 4 + r] - 1);
    printf("\n");
  }
  // ---- 32x32x2: use only the k=0 half (lanes 0..31) nonzero in A, all ones in B -> D = A lane; and vice versa
  for (int half = 0; half < 2; ++half) {
    for (int l = 0; l < 64; ++l) { ha[l] = ((l >> 5) == half) ? (l + 1) : 0; hb[l] = ((l >> 5) == half) ? 1 : 0; }
    hipMemcpy(a, ha.data(), 256, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice);
    k32<<<1, 64>>>(a, b, d); hipMemcpy(hd.data(), d, 64 * 16 * 4, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) { ha[l] = ((l >> 5) == half) ? 1 : 0; hb[l] = ((l >> 5) == half) ? (l + 1) : 0; }
    hipMemcpy(a, ha.data(), 256, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice);
    k32<<<1, 64>>>(a, b, d); hipMemcpy(hd2.data(), d, 64 * 16 * 4, hipMemcpyDeviceToHost);
    printf("32x32x2 k-half %d: D[lane][reg] <- (A lane, B lane); expect A lane = row + 32*half with row=(reg&3)+8*(reg>>2)+4*(lane>>5), B lane = (lane&31) + 32*half\n", half);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 16; ++r) {
        int la = (int)hd[l * 16 + r] - 1, lb = (int)hd2[l * 16 + r] - 1;
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        if (la != row + 32 * half || lb != (l & 31) + 32 * half) { if (bad < 8) printf("  MISMATCH lane %d reg %d: A %d B %d\n", l, r, la, lb); ++bad; }
      }
    printf("  mismatches: %d\n", bad);
  }
  // ---- 16x16x4: k-slot g = lane>>4
  for (int g = 0; g < 4; ++g) {
    for (int l = 0; l < 64; ++l) { ha[l] = ((l >> 4) == g) ? (l + 1) : 0; hb[l] = ((l >> 4) == g) ? 1 : 0; }
    hipMemcpy(a, ha.data(), 256, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice);
    k16<<<1, 64>>>(a, b, d); hipMemcpy(hd.data(), d, 64 * 4 * 4, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) { ha[l] = ((l >> 4) == g) ? 1 : 0; hb[l] = ((l >> 4) == g) ? (l + 1) : 0; }
    hipMemcpy(a, ha.data(), 256, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice);
    k16<<<1, 64>>>(a, b, d); hipMemcpy(hd2.data(), d, 64 * 4 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 4; ++r) {
        int la = (int)hd[l * 4 + r] - 1, lb = (int)hd2[l * 4 + r] - 1;
        int row = 4 * (l >> 4) + r;
        if (la != row + 16 * g || lb != (l & 15) + 16 * g) { if (bad < 4) printf("  16x16x4 MISMATCH lane %d reg %d: A %d B %d\n", l, r, la, lb); ++bad; }
      }
    printf("16x16x4 k-slot %d mismatches vs (row=4*(lane>>4)+reg, col=lane&15): %d\n", g, bad);
  }
  return 0;
}
