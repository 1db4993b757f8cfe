#include "common.h"
#include <string>

namespace srh {
static thread_local std::string g_last_error;
void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
}
const std::string& last_error() { return g_last_error; }

int cu_count() {
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev] = n;
  }
  return cached[dev];
}

srh_status_t check_fetch_args(const srh_batch_fetch_args_t* in, srh_batch_fetch_args_t& out) {
  SRH_REQUIRE(in, "batch_fetch: null argument");
  SRH_REQUIRE(in->d_epoch_u && in->d_epoch_i && in->d_epoch_j && in->d_cursor && in->d_stage_u && in->d_stage_i &&
                  in->d_stage_j && in->d_meta,
              "batch_fetch: null argument");
  const bool uq = in->d_epoch_uniq_u != nullptr;
  SRH_REQUIRE(!uq || (in->d_epoch_uniq_i && in->d_n_uniq_u && in->d_n_uniq_i && in->d_stage_uniq_u && in->d_stage_uniq_i),
              "batch_fetch: unique-id arrays must be given together");
  SRH_REQUIRE(in->n_edges > 0 && in->batch_size > 0, "batch_fetch: bad sizes");
  SRH_REQUIRE(!in->d_adam_coef || (in->adam_lr > 0.f && in->adam_beta1 >= 0.f && in->adam_beta1 < 1.f &&
                                   in->adam_beta2 >= 0.f && in->adam_beta2 < 1.f),
              "batch_fetch: d_adam_coef needs adam_lr > 0 and betas in [0, 1)");
  out = *in;
  if (!uq) { out.d_stage_cat = nullptr; out.d_n_cat = nullptr; }
  return SRH_OK;
}
}  // namespace srh

extern "C" {
int32_t srh_abi_version(void) { return SRH_ABI_VERSION; }
const char* srh_last_error_string(void) { return srh::last_error().c_str(); }
int32_t srh_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}
}
