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
