// Probe: the wave reductions of the kernels (DPP row steps + row_bcast + readlane, ey_mfma32.hip wsum) compiled with the
// allocator of the `make spill` builds (-mllvm -vgpr-regalloc=fast) and without: do they still sum?
//   hipcc -O3 --offload-arch=gfx950 [-mllvm -vgpr-regalloc=fast] tools/dpp_fastra_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_get(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xF, false));
}
__device__ __forceinline__ float wsum(float v) {
  v += dpp_get<0xB1, 0xF>(v);
  v += dpp_get<0x4E, 0xF>(v);
  v += dpp_get<0x141, 0xF>(v);
  v += dpp_get<0x140, 0xF>(v);
  v += dpp_get<0x142, 0xA>(v);
  v += dpp_get<0x143, 0xC>(v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__global__ void k(const float* x, float* out, int n) {
  const int lane = threadIdx.x & 63;
  float acc[8];
  for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < 8; ++j) acc[j] += x[(i * 8 + j) * 64 + lane];
  float r[8];
  for (int j = 0; j < 8; ++j) r[j] = wsum(acc[j]);
  // a sum under a per-lane condition, as the kernels' `if (counts) kin += v * v`
  float kin = 0.0f;
  for (int j = 0; j < 8; ++j) if ((lane >> j) & 1) kin += acc[j] * acc[j];
  kin = wsum(kin);
  if (lane == 0) { for (int j = 0; j < 8; ++j) out[j] = r[j]; out[8] = kin; }
}
int main() {
  const int n = 3;
  float hx[3 * 8 * 64], *dx, *dout, ho[9];
  for (int i = 0; i < n * 8 * 64; ++i) hx[i] = (float)((i * 7) % 13) - 6.0f;
  (void)hipMalloc(&dx, sizeof(hx)); (void)hipMalloc(&dout, sizeof(ho));
  (void)hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dout, n);
  (void)hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
  int bad = 0;
  double kin = 0;
  for (int j = 0; j < 8; ++j) {
    double s = 0;
    for (int lane = 0; lane < 64; ++lane) {
      double a = 0;
      for (int i = 0; i < n; ++i) a += hx[(i * 8 + j) * 64 + lane];
      s += a;
      if ((lane >> j) & 1) kin += a * a;
    }
    printf("sum %d: device %.1f host %.1f%s\n", j, ho[j], s, ho[j] == (float)s ? "" : "   <-- WRONG");
    bad += ho[j] != (float)s;
  }
  printf("conditional sum: device %.1f host %.1f%s\n", ho[8], kin, ho[8] == (float)kin ? "" : "   <-- WRONG");
  bad += ho[8] != (float)kin;
  printf(bad ? "FAIL\n" : "ok\n");
  return bad != 0;
}
