// Shared helpers for libselfrec_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include "../../include/selfrec_hip.h"

namespace srh {

void set_error(const char* fmt, ...);

#define SRH_REQUIRE(cond, ...)                    \
  do {                                            \
    if (!(cond)) {                                \
      ::srh::set_error(__VA_ARGS__);              \
      return SRH_ERR_INVALID_ARG;                 \
    }                                             \
  } while (0)

#define SRH_HIP(call)                                                              \
  do {                                                                             \
    hipError_t e_ = (call);                                                        \
    if (e_ != hipSuccess) {                                                        \
      ::srh::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),      \
                       __FILE__, __LINE__);                                        \
      return SRH_ERR_HIP;                                                          \
    }                                                                              \
  } while (0)

// Every launch is followed by this: catches bad configurations without synchronising.
#define SRH_LAUNCH_CHECK() SRH_HIP(hipGetLastError())

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int kWave = 64;  // gfx950 wavefront

// lanes per embedding row when each lane owns one float4 of it
inline bool dim_supported(int d) { return d == 32 || d == 64 || d == 128 || d == 256; }

}  // namespace srh

#ifdef __HIPCC__
namespace srh {

__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
__device__ __forceinline__ float4 f4_fma(float s, float4 x, float4 acc) {
  return make_float4(fmaf(s, x.x, acc.x), fmaf(s, x.y, acc.y), fmaf(s, x.z, acc.z), fmaf(s, x.w, acc.w));
}
__device__ __forceinline__ float4 f4_scale(float4 a, float s) {
  return make_float4(a.x * s, a.y * s, a.z * s, a.w * s);
}
__device__ __forceinline__ float f4_dot(float4 a, float4 b) {
  return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
}
__device__ __forceinline__ float4 f4_shfl_xor(float4 v, int mask) {
  return make_float4(__shfl_xor(v.x, mask), __shfl_xor(v.y, mask), __shfl_xor(v.z, mask), __shfl_xor(v.w, mask));
}
// Sum over the LPR lanes that share one embedding row (LPR a power of two <= 64).
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int m = 1; m < LPR; m <<= 1) v += __shfl_xor(v, m);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
  return v;
}
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
  return v;
}
__device__ __forceinline__ float sgnf(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }

}  // namespace srh
#endif
