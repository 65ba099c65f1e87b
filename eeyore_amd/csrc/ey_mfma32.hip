// MFMA kernel family for MLP(d0-32-32-dK) in f32 -- placeholder until the fused trajectory kernel lands.
#include "ey_common.h"

bool ey_mfma32_supports(const ey_plan*) { return false; }
int ey_mfma32_set_data(ey_plan*, hipStream_t) { return EY_OK; }
int ey_mfma32_hmc(ey_plan*, void*, void*, void*, const void*, const void*, double, const void*, int, const void*,
                  int64_t, uint64_t, uint64_t, uint64_t, uint32_t, void*, void*, void*, void*, hipStream_t) {
  EY_FAIL(EY_ERR_UNSUPPORTED, "mfma32 path not built");
}
int ey_mfma32_log_target_grad(ey_plan*, const void*, const void*, int64_t, void*, void*, hipStream_t) {
  EY_FAIL(EY_ERR_UNSUPPORTED, "mfma32 path not built");
}
