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

// compute units of the current device (cached per device; 256 on MI355X)
int cu_count();

// the checks srh_batch_fetch promises; `out` = the args as the kernels take them (optional groups normalised)
srh_status_t check_fetch_args(const srh_batch_fetch_args_t* in, srh_batch_fetch_args_t& out);

}  // namespace srh

#ifdef __HIPCC__
namespace srh {

// Stages batch *cursor[0] of the epoch arrays into fixed buffers, publishes its sizes and stamps the
// activity marks with the current optimiser step *cursor[1].  Read-only on the cursor, so any number of
// workgroups can share the copy; the cursor is advanced at the END of the step (Adam's reset pass / zero_rows_kernel /
// cursor_advance_kernel), after the last kernel that reads it.  `block` of `n_blocks` 256-thread workgroups: the body
// of batch_fetch_kernel (optim.hip) and of the fetch workgroups riding on an SpMM launch (spmm.hip).
constexpr int kFetchBlocks = 8;
__device__ __forceinline__ void batch_fetch_body(const srh_batch_fetch_args_t& f, const int block, const int n_blocks) {
  const int64_t b = f.d_cursor[0];
  const int32_t stamp = (int32_t)f.d_cursor[1];
  const int64_t bs = f.batch_size, ptr = b * bs;
  // (two epochs back to back: batch b >= half_batches is batch b - half_batches of the second one)
  const int64_t in_epoch = ((f.half_batches > 0 && b >= f.half_batches) ? b - f.half_batches : b) * bs;
  const int64_t rows = (in_epoch >= f.n_edges) ? 0 : ((in_epoch + bs < f.n_edges) ? bs : f.n_edges - in_epoch);
  const int64_t tid = (int64_t)block * 256 + threadIdx.x, nth = (int64_t)n_blocks * 256;
  for (int64_t i = tid; i < rows; i += nth) {
    const int32_t u = f.d_epoch_u[ptr + i], p = f.d_epoch_i[ptr + i], n = f.d_epoch_j[ptr + i];
    f.d_stage_u[i] = u; f.d_stage_i[i] = p; f.d_stage_j[i] = n;
    if (f.d_row_mark) {
      f.d_row_mark[u] = stamp; f.d_row_mark[f.mark_item_offset + p] = stamp; f.d_row_mark[f.mark_item_offset + n] = stamp;
    }
  }
  int32_t a = 0, c = 0;
  if (f.d_epoch_uniq_u && rows > 0) {
    a = f.d_n_uniq_u[b];
    c = f.d_n_uniq_i[b];
    for (int64_t i = tid; i < a; i += nth) f.d_stage_uniq_u[i] = f.d_epoch_uniq_u[b * bs + i];
    for (int64_t i = tid; i < c; i += nth) f.d_stage_uniq_i[i] = f.d_epoch_uniq_i[b * bs + i];
    if (f.d_stage_cat) {   // [unique users ; unique items] as one index list (SGL.py:120-125 concatenates the two sides)
      for (int64_t i = tid; i < a; i += nth) f.d_stage_cat[i] = f.d_epoch_uniq_u[b * bs + i];
      for (int64_t i = tid; i < c; i += nth) f.d_stage_cat[a + i] = f.d_epoch_uniq_i[b * bs + i] + f.cat_item_offset;
    }
  }
  if (f.d_n_cat && tid == 0) *f.d_n_cat = a + c;
  if (f.d_zero4 && tid < 4) f.d_zero4[tid] = 0.0;        // the step's loss accumulators
  if (f.d_now && tid < 2) f.d_now[tid] = f.d_cursor[tid];
  if (f.d_adam_coef && tid == 0) {
    // torch: bias_correction = 1 - beta ** step (python double), step_size = lr / bc1, denom = sqrt(v) / sqrt(bc2) + eps
    // (the same expressions as adam_kernel, optim.hip -- same bits)
    const double t = (double)f.d_cursor[1];
    f.d_adam_coef[0] = (float)((double)f.adam_lr / (1.0 - pow((double)f.adam_beta1, t)));
    f.d_adam_coef[1] = (float)sqrt(1.0 - pow((double)f.adam_beta2, t));
  }
  if (tid == 0) {
    f.d_meta[0] = (int32_t)rows;
    f.d_meta[1] = a;
    f.d_meta[2] = c;
    f.d_meta[3] = (int32_t)b;
  }
}

__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
// 16-byte store of a streamed output.  WT = write-through (`sc1`): the bytes go to the fabric as they are produced and the
// line is not left dirty in this XCD's L2 -- a kernel that ends with tens of MB of dirty lines pays their write-back at
// its boundary (MI355X_MICROARCH.md price list, row "boundary": + B / 6 TB/s), and nothing on this XCD reads them again.
template <bool WT>
__device__ __forceinline__ void st_f4(float4* p, float4 v) {
  if constexpr (WT) {
    typedef float fx4_t __attribute__((ext_vector_type(4)));
    const fx4_t x = {v.x, v.y, v.z, v.w};
    // (s_nop 1 INSIDE the statement: a 16-byte store reads its data registers after issue, and the compiler pads no hazards
    // of an asm statement -- without it the next instruction may overwrite them first: cdna_hip_programming.md 5.7)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
  } else {
    *p = v;
  }
}
// One element of torch.optim.Adam's step (no weight decay, no amsgrad): the ONE definition both the stand-alone pass
// (adam_kernel, optim.hip) and the SpMM epilogue (SRH_EPI_ADAM, spmm.hip) inline, so that they round alike.
//   step_size = lr / (1 - beta1^t), bc2_sqrt = sqrt(1 - beta2^t)
__device__ __forceinline__ void adam_element(float& m, float& v, float& p, const float g, const float b1, const float omb1,
                                             const float b2, const float omb2, const float step_size, const float bc2_sqrt,
                                             const float eps) {
#pragma clang fp contract(off)    // the fused multiply-adds are the three written here, in every caller
  m = fmaf(m, b1, omb1 * g);
  v = fmaf(v, b2, omb2 * (g * g));
  p = fmaf(-step_size, m / (sqrtf(v) / bc2_sqrt + eps), p);
}
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
