// (a-9) Dense Adam and the small streaming utilities of the fused step.
// Replaces torch.optim.Adam(model.parameters(), lr).step() (reference XSimGCL.py:25,37):
// betas/eps as passed, no weight decay, bias-corrected, one launch over the whole (N, d)
// table instead of torch's ~10 elementwise launches.  HBM-bound:
//   algorithmic bytes = 7 * n_elem * 4  (read p, g, m, v; write p, m, v).
#include "common.h"

// Build-time switch for an A/B (tools/spmm_lab/build_alt.sh): SRH_ADAM_WT 1 (default) = p, m, v leave with write-through
// (`sc1`) stores -- 53 MB per step that the next launch's gathers read from other XCDs, not left dirty for the kernel
// boundary to write back.  Together with the InfoNCE partials (losses.hip: SRH_NCE_WT): 0.2813 -> 0.2776 ms per step in a
// same-box A/B, each alone -1.2 us (profiles/r04_c_write_through_ab.txt).
#ifndef SRH_ADAM_WT
#define SRH_ADAM_WT 1
#endif

namespace {
using namespace srh;
constexpr bool kAdamWT = SRH_ADAM_WT != 0;

// Optional fused reset (the engine's step): tables in `clr` get the rows whose activity mark equals this step's stamp
// zeroed in the same pass -- the batch-sparse gradient buffers (gF, gCL, ...) hold non-zeros only there -- and
// `cursor` (batch no, step) is advanced by block 0.  The step itself is read from d_step, which in that mode is
// batch_fetch's COPY of the cursor: nobody reads `cursor` while this kernel runs.
struct AdamClear {
  float4* table[SRH_MAX_ADAM_CLEAR];
  int32_t n;
  const int32_t* mark;
  int32_t lpr_shift;      // log2(float4 per table row)
  int64_t* cursor;
};

__global__ __launch_bounds__(256) void adam_kernel(float4* __restrict__ p, const float4* g /* may alias a table to clear */,
                                                   float4* __restrict__ m, float4* __restrict__ v, int64_t n4,
                                                   int64_t step, const int64_t* __restrict__ d_step, float lr,
                                                   float b1, float b2, float eps, AdamClear clr) {
  const int64_t t = d_step ? *d_step : step;
  if (clr.cursor && blockIdx.x == 0 && threadIdx.x == 0) { clr.cursor[0] += 1; clr.cursor[1] += 1; }
  // torch: bias_correction = 1 - beta ** step (python double), step_size = lr / bc1,
  //        denom = sqrt(v) / sqrt(bc2) + eps, p -= step_size * m / denom
  const double bc1 = 1.0 - pow((double)b1, (double)t);
  const double bc2 = 1.0 - pow((double)b2, (double)t);
  const float step_size = (float)((double)lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const float4 gg = g[i];
    float4 mm = m[i], vv = v[i], pp = p[i];
#define SRH_ADAM_LANE(c) adam_element(mm.c, vv.c, pp.c, gg.c, b1, omb1, b2, omb2, step_size, bc2_sqrt, eps);
    SRH_ADAM_LANE(x) SRH_ADAM_LANE(y) SRH_ADAM_LANE(z) SRH_ADAM_LANE(w)
#undef SRH_ADAM_LANE
    st_f4<kAdamWT>(m + i, mm); st_f4<kAdamWT>(v + i, vv); st_f4<kAdamWT>(p + i, pp);
    if (clr.n > 0 && clr.mark[i >> clr.lpr_shift] == (int32_t)t) {
      for (int k = 0; k < clr.n; ++k) clr.table[k][i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

__global__ __launch_bounds__(256) void axpby_kernel(float a, const float4* __restrict__ x, float b,
                                                    float4* __restrict__ y, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const float4 xv = x[i];
    float4 o;
    if (b == 0.f) {
      o = make_float4(a * xv.x, a * xv.y, a * xv.z, a * xv.w);
    } else {
      const float4 yv = y[i];
      o = make_float4(a * xv.x + b * yv.x, a * xv.y + b * yv.y, a * xv.z + b * yv.z, a * xv.w + b * yv.w);
    }
    y[i] = o;
  }
}

__global__ __launch_bounds__(256) void batch_fetch_kernel(srh_batch_fetch_args_t f) {
  srh::batch_fetch_body(f, (int)blockIdx.x, (int)gridDim.x);
}

__global__ void cursor_advance_kernel(int64_t* cursor) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { cursor[0] += 1; cursor[1] += 1; }
}

struct ZeroList {
  float* table[SRH_MAX_ZERO_LISTS];
  const int32_t* idx[SRH_MAX_ZERO_LISTS];
  const int32_t* d_n[SRH_MAX_ZERO_LISTS];
  int32_t n_max[SRH_MAX_ZERO_LISTS];
  int32_t offset[SRH_MAX_ZERO_LISTS];
  int32_t n_lists;
  int64_t* cursor;     // optional: advanced by one batch / one step when this (last) kernel of the step runs
};

// zero the listed rows of (.., d) tables: the sparse counterpart of a dense memset for
// gradient buffers that only ever receive O(batch) non-zero rows
__global__ __launch_bounds__(256) void zero_rows_kernel(ZeroList z, int lpr) {
  if (z.cursor && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { z.cursor[0] += 1; z.cursor[1] += 1; }
  const int list = blockIdx.y;
  if (list >= z.n_lists) return;
  const int n = z.d_n[list] ? min(*z.d_n[list], z.n_max[list]) : z.n_max[list];
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t r = t / lpr;
  if (r >= n) return;
  const int64_t row = (int64_t)z.idx[list][r] + z.offset[list];
  reinterpret_cast<float4*>(z.table[list])[row * lpr + (t % lpr)] = make_float4(0.f, 0.f, 0.f, 0.f);
}
}  // namespace

extern "C" {

static srh_status_t adam_launch(float* d_param, const float* d_grad, float* d_m, float* d_v, int64_t n_elem,
                                int64_t step, const int64_t* d_step, float lr, float beta1, float beta2, float eps,
                                const AdamClear& clr, void* stream) {
  SRH_REQUIRE(d_param && d_grad && d_m && d_v, "adam_step: null argument");
  SRH_REQUIRE(n_elem > 0 && n_elem % 4 == 0, "adam_step: n_elem must be a positive multiple of 4");
  SRH_REQUIRE(d_step || step >= 1, "adam_step: step is 1-based");
  const int64_t n4 = n_elem / 4;
  const int blocks = (int)std::min<int64_t>((n4 + 255) / 256, 256 * 16);
  adam_kernel<<<blocks, 256, 0, srh::as_stream(stream)>>>(
      reinterpret_cast<float4*>(d_param), reinterpret_cast<const float4*>(d_grad), reinterpret_cast<float4*>(d_m),
      reinterpret_cast<float4*>(d_v), n4, step, d_step, lr, beta1, beta2, eps, clr);
  SRH_LAUNCH_CHECK();
  return SRH_OK;
}

srh_status_t srh_adam_step(float* d_param, const float* d_grad, float* d_m, float* d_v, int64_t n_elem,
                           int64_t step, const int64_t* d_step, float lr, float beta1, float beta2, float eps,
                           void* stream) {
  return adam_launch(d_param, d_grad, d_m, d_v, n_elem, step, d_step, lr, beta1, beta2, eps, AdamClear{}, stream);
}

srh_status_t srh_adam_step_reset(float* d_param, const float* d_grad, float* d_m, float* d_v, int64_t n_rows,
                                 int32_t d, const int64_t* d_step, float lr, float beta1, float beta2, float eps,
                                 const int32_t* d_row_mark, int32_t n_clear, float* const* d_clear_tables,
                                 int64_t* d_cursor_advance, void* stream) {
  SRH_REQUIRE(d_step && d_row_mark, "adam_step_reset: the step and the row marks live in device memory");
  SRH_REQUIRE(n_rows > 0 && d > 0 && d % 4 == 0, "adam_step_reset: bad shape");
  SRH_REQUIRE(n_clear >= 0 && n_clear <= SRH_MAX_ADAM_CLEAR && (n_clear == 0 || d_clear_tables),
              "adam_step_reset: at most %d tables to clear", SRH_MAX_ADAM_CLEAR);
  AdamClear clr{};
  for (int k = 0; k < n_clear; ++k) {
    SRH_REQUIRE(d_clear_tables[k], "adam_step_reset: null table %d", k);
    clr.table[k] = reinterpret_cast<float4*>(d_clear_tables[k]);
  }
  const int lpr = d / 4;
  SRH_REQUIRE((lpr & (lpr - 1)) == 0, "adam_step_reset: d = %d (rows of a power-of-two number of float4)", d);
  int shift = 0;
  while ((1 << shift) < lpr) ++shift;
  clr.n = n_clear; clr.mark = d_row_mark; clr.lpr_shift = shift; clr.cursor = d_cursor_advance;
  return adam_launch(d_param, d_grad, d_m, d_v, n_rows * d, 0, d_step, lr, beta1, beta2, eps, clr, stream);
}

srh_status_t srh_axpby(float a, const float* d_x, float b, float* d_y, int64_t n_elem, void* stream) {
  SRH_REQUIRE(d_x && d_y, "axpby: null argument");
  SRH_REQUIRE(n_elem > 0 && n_elem % 4 == 0, "axpby: n_elem must be a positive multiple of 4");
  const int64_t n4 = n_elem / 4;
  const int blocks = (int)std::min<int64_t>((n4 + 255) / 256, 256 * 16);
  axpby_kernel<<<blocks, 256, 0, srh::as_stream(stream)>>>(a, reinterpret_cast<const float4*>(d_x), b,
                                                          reinterpret_cast<float4*>(d_y), n4);
  SRH_LAUNCH_CHECK();
  return SRH_OK;
}

srh_status_t srh_batch_fetch(const srh_batch_fetch_args_t* args, void* stream) {
  srh_batch_fetch_args_t f;
  if (srh_status_t rc = srh::check_fetch_args(args, f)) return rc;
  batch_fetch_kernel<<<srh::kFetchBlocks, 256, 0, srh::as_stream(stream)>>>(f);
  SRH_LAUNCH_CHECK();
  return SRH_OK;
}

srh_status_t srh_cursor_advance(int64_t* d_cursor, void* stream) {
  SRH_REQUIRE(d_cursor, "cursor_advance: null argument");
  cursor_advance_kernel<<<1, 64, 0, srh::as_stream(stream)>>>(d_cursor);
  SRH_LAUNCH_CHECK();
  return SRH_OK;
}

srh_status_t srh_zero_rows(int32_t n_lists, float* const* d_tables, const int32_t* const* d_idx,
                           const int32_t* const* d_counts, const int32_t* n_max, const int32_t* row_offset,
                           int32_t d, int64_t* d_cursor_advance, void* stream) {
  SRH_REQUIRE(n_lists >= 1 && n_lists <= SRH_MAX_ZERO_LISTS, "zero_rows: 1..%d lists", SRH_MAX_ZERO_LISTS);
  SRH_REQUIRE(d_tables && d_idx && d_counts && n_max && row_offset, "zero_rows: null argument");
  SRH_REQUIRE(d > 0 && d % 4 == 0, "zero_rows: d must be a positive multiple of 4");
  ZeroList z{};
  int32_t most = 0;
  for (int k = 0; k < n_lists; ++k) {
    SRH_REQUIRE(d_tables[k] && d_idx[k] && n_max[k] >= 0, "zero_rows: bad list %d", k);
    z.table[k] = d_tables[k]; z.idx[k] = d_idx[k]; z.d_n[k] = d_counts[k];
    z.n_max[k] = n_max[k]; z.offset[k] = row_offset[k];
    most = std::max(most, n_max[k]);
  }
  z.n_lists = n_lists;
  z.cursor = d_cursor_advance;
  if (most == 0) return d_cursor_advance ? srh_cursor_advance(d_cursor_advance, stream) : SRH_OK;
  const int lpr = d / 4;
  dim3 grid((unsigned)(((int64_t)most * lpr + 255) / 256), (unsigned)n_lists);
  zero_rows_kernel<<<grid, 256, 0, srh::as_stream(stream)>>>(z, lpr);
  SRH_LAUNCH_CHECK();
  return SRH_OK;
}

}  // extern "C"
