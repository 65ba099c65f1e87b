// Which XCD runs workgroup `id` of a 1-D / 3-D grid?  s_getreg_b32 XCC_ID per workgroup, printed as the pattern over the first
// ids and as a histogram of (id % 8 == xcc) agreement.  hipcc --offload-arch=gfx950 -O2 -o tools/xcd_probe tools/xcd_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned* out, int spin) {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  const unsigned id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  float a = threadIdx.x;
  for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;  // keep the workgroup resident for a while
  if (threadIdx.x == 0) out[id] = (x & 15u) | ((hw >> 8 & 15u) << 8) | ((a > 1e30f) << 31);  // CU id bits 11:8
}
int main() {
  for (int pass = 0; pass < 3; ++pass) {
    dim3 grid = pass == 0 ? dim3(4096) : pass == 1 ? dim3(6, 1, 4096) : dim3(1, 8, 4096);
    const unsigned total = grid.x * grid.y * grid.z;
    unsigned* d;
    hipMalloc(&d, total * 4);
    hipLaunchKernelGGL(k, grid, dim3(256), 0, 0, d, 20000);
    std::vector<unsigned> h(total);
    hipMemcpy(h.data(), d, total * 4, hipMemcpyDeviceToHost);
    unsigned agree = 0, hist[16] = {0};
    for (unsigned i = 0; i < total; ++i) { agree += (h[i] & 15u) == (i & 7u); hist[h[i] & 15u]++; }
    printf("grid (%u, %u, %u): xcc of ids 0..31:", grid.x, grid.y, grid.z);
    for (int i = 0; i < 32; ++i) printf(" %u", h[i] & 15u);
    printf("\n  ids 768..799:");
    for (int i = 768; i < 800; ++i) printf(" %u", h[i] & 15u);
    printf("\n  xcc == id %% 8 for %u of %u workgroups; per XCC:", agree, total);
    for (int i = 0; i < 8; ++i) printf(" %u", hist[i]);
    printf("\n");
    hipFree(d);
  }
  return 0;
}
