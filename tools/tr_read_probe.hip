#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
// LDS image: [row 0..15][col 0..127] of 16-bit values, value = row * 256 + col; 256-byte rows, no swizzle
__global__ void k(unsigned short* out, int r0, int c0) {
  __shared__ __attribute__((aligned(16))) unsigned short img[16 * 128];
  for (int i = threadIdx.x; i < 16 * 128; i += 64) img[i] = (unsigned short)((i / 128) * 256 + (i % 128));
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  // lane 4q+p of a 16-lane group supplies the address of row q, columns 4p .. 4p+3 of the group's block;
  // block of group g: rows r0 + 4*(g&1) .., columns c0 + 16*(g>>1)
  const int row = r0 + 4 * (g & 1) + q, col = c0 + 16 * (g >> 1) + 4 * p;
  typedef __attribute__((address_space(3))) s16x4 lds_v;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(img + row * 128 + col));
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (unsigned short)v[e];
}
int main() {
  unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
  k<<<1, 64>>>(d, 0, 0);
  unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int e = 0; e < 4; ++e) printf(" (r%d,c%3d)", h[l * 4 + e] >> 8, h[l * 4 + e] & 255);
    printf("\n");
  }
  return 0;
}
