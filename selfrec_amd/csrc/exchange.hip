// (e) Column-sharded tables: the batch-row exchange around the loss section.
//
// With every (N, d) table split by COLUMNS over the G ranks (rank r keeps columns [r*dl, (r+1)*dl),
// dl = d / G), the propagation layers of LightGCN.py:72 / XSimGCL.py:88 need no communication at all
// -- a sparse product is independent per column -- and the only rows any rank ever needs whole are the
// O(batch) rows the losses read (XSimGCL.py:30-33,45-50; loss_torch.py:6-10,18-22,35-50).  One step
// therefore has ONE collective: an all-gather of those rows' slices.
//
//   srh_batch_pack     rows listed by the staged batch (u | i | j | unique users | unique items, B slots
//                      each) of up to 4 local (N, dl) tables -> send buffer (n_tables, 5B, dl)
//   (all-gather)       recv (G, n_tables, 5B, dl)                                     -- RCCL, host side
//   srh_batch_unpack   recv -> "compact" tables (5B, d): row k = slot k of the batch lists, whole rows;
//                      also clears the compact gradient tables' live rows
//   (losses)           the unchanged BPR / InfoNCE kernels run on the compact tables with the constant
//                      index lists slot -> slot; gradients land in compact (5B, d) tables
//   srh_batch_scatter  this rank's column slice of each compact gradient row is added to the node's row
//                      of the local (N, dl) gradient table
//
// Bound: latency (a few MB, O(batch)); HBM-bound byte moves, no arithmetic.
#include "common.h"

namespace {

using namespace srh;

constexpr int kSegs = 5;

struct Lists {
  const int32_t* idx[kSegs];
  const int32_t* count[kSegs];
  int32_t B;
};

struct Ptrs4 {
  const float* src[SRH_MAX_EXCHANGE];
  float* dst[SRH_MAX_EXCHANGE];
};

__device__ __forceinline__ int seg_count(const Lists& l, int s) {
  return l.count[s] ? min(*l.count[s], l.B) : l.B;
}

// one thread per float4 of the send buffer
__global__ __launch_bounds__(256) void pack_kernel(Lists l, Ptrs4 t, int n_tables, int q_per_row /* dl/4 */,
                                                   float4* __restrict__ send, int32_t* __restrict__ cat_idx,
                                                   int32_t* __restrict__ n_cat) {
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t rows = (int64_t)kSegs * l.B;
  if (cat_idx) {          // compact slots of [unique users ; unique items] as ONE list (SGL.py:120-125)
    const int a = seg_count(l, 3), c = seg_count(l, 4);
    if (tid < a) cat_idx[tid] = 3 * l.B + (int)tid;
    else if (tid < a + c) cat_idx[tid] = 4 * l.B + (int)(tid - a);
    if (tid == 0 && n_cat) *n_cat = a + c;
  }
  const int64_t total = rows * q_per_row * n_tables;
  if (tid >= total) return;
  const int q = (int)(tid % q_per_row);
  const int64_t rk = tid / q_per_row;
  const int k = (int)(rk % rows), tb = (int)(rk / rows);
  const int s = k / l.B, pos = k % l.B;
  float4 v = f4_zero();
  if (pos < seg_count(l, s)) {
    const int64_t node = l.idx[s][pos];
    v = reinterpret_cast<const float4*>(t.src[tb])[node * q_per_row + q];
  }
  send[tid] = v;
}

// one thread per float4 of a compact row, over tables then gradient tables
__global__ __launch_bounds__(256) void unpack_kernel(Lists l, Ptrs4 t, int n_tables, int n_grads, int world,
                                                     int q_per_row /* dl/4 */, const float4* __restrict__ recv) {
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t rows = (int64_t)kSegs * l.B;
  const int qf = q_per_row * world;                  // float4 per whole row
  const int64_t per_table = rows * qf;
  if (tid >= per_table * (n_tables + n_grads)) return;
  const int tb = (int)(tid / per_table);
  const int64_t rem = tid % per_table;
  const int k = (int)(rem / qf), q = (int)(rem % qf);
  const int s = k / l.B, pos = k % l.B;
  if (pos >= seg_count(l, s)) return;
  if (tb >= n_tables) {                              // gradient tables start every step at zero
    reinterpret_cast<float4*>(t.dst[tb])[rem] = f4_zero();
    return;
  }
  const int g = q / q_per_row, qq = q % q_per_row;
  reinterpret_cast<float4*>(t.dst[tb])[rem] = recv[(((int64_t)g * n_tables + tb) * rows + k) * q_per_row + qq];
}

// one thread per (pair, compact row, local column)
__global__ __launch_bounds__(256) void scatter_kernel(Lists l, Ptrs4 t, int n_pairs, int d_full, int col0, int dl) {
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t rows = (int64_t)kSegs * l.B;
  if (tid >= rows * dl * n_pairs) return;
  const int c = (int)(tid % dl);
  const int64_t rk = tid / dl;
  const int k = (int)(rk % rows), p = (int)(rk / rows);
  const int s = k / l.B, pos = k % l.B;
  if (pos >= seg_count(l, s)) return;
  const float v = t.src[p][(int64_t)k * d_full + col0 + c];
  if (v == 0.f) return;                              // (most segments of most gradient tables are untouched)
  const int64_t node = l.idx[s][pos];
  atomicAdd(t.dst[p] + node * dl + c, v);            // the same node can sit in several slots
}

srh_status_t take_lists(const srh_batch_lists_t* in, Lists& out, const char* who) {
  if (!in) { srh::set_error("%s: null lists", who); return SRH_ERR_INVALID_ARG; }
  if (in->B <= 0) { srh::set_error("%s: bad batch size %d", who, in->B); return SRH_ERR_INVALID_ARG; }
  for (int s = 0; s < kSegs; ++s) {
    if (!in->d_idx[s]) { srh::set_error("%s: null index list %d", who, s); return SRH_ERR_INVALID_ARG; }
    out.idx[s] = in->d_idx[s];
    out.count[s] = in->d_count[s];
  }
  out.B = in->B;
  return SRH_OK;
}

}  // namespace

extern "C" {

srh_status_t srh_batch_pack(const srh_batch_lists_t* lists, int32_t n_tables, const float* const* d_tables, int32_t dl,
                            float* d_send, int32_t* d_cat_idx, int32_t* d_n_cat, void* stream) {
  Lists l;
  if (srh_status_t st = take_lists(lists, l, "batch_pack")) return st;
  SRH_REQUIRE(n_tables >= 1 && n_tables <= SRH_MAX_EXCHANGE && d_tables && d_send, "batch_pack: 1..%d tables", SRH_MAX_EXCHANGE);
  SRH_REQUIRE(dl > 0 && dl % 4 == 0, "batch_pack: the column slice must be a positive multiple of 4 (got %d)", dl);
  Ptrs4 t{};
  for (int k = 0; k < n_tables; ++k) {
    SRH_REQUIRE(d_tables[k], "batch_pack: null table %d", k);
    t.src[k] = d_tables[k];
  }
  const int64_t total = (int64_t)kSegs * l.B * (dl / 4) * n_tables;
  const int64_t threads = total > 2 * (int64_t)l.B ? total : 2 * (int64_t)l.B;
  pack_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, srh::as_stream(stream)>>>(
      l, t, n_tables, dl / 4, reinterpret_cast<float4*>(d_send), d_cat_idx, d_n_cat);
  SRH_LAUNCH_CHECK();
  return SRH_OK;
}

srh_status_t srh_batch_unpack(const srh_batch_lists_t* lists, int32_t n_tables, int32_t world, int32_t dl,
                              const float* d_recv, float* const* d_compact, int32_t n_grads,
                              float* const* d_compact_grads, void* stream) {
  Lists l;
  if (srh_status_t st = take_lists(lists, l, "batch_unpack")) return st;
  SRH_REQUIRE(n_tables >= 1 && n_grads >= 0 && n_tables + n_grads <= SRH_MAX_EXCHANGE && d_compact && d_recv &&
                  (n_grads == 0 || d_compact_grads),
              "batch_unpack: at most %d tables + gradient tables", SRH_MAX_EXCHANGE);
  SRH_REQUIRE(world >= 1 && dl > 0 && dl % 4 == 0, "batch_unpack: bad world / slice (%d, %d)", world, dl);
  Ptrs4 t{};
  for (int k = 0; k < n_tables; ++k) {
    SRH_REQUIRE(d_compact[k], "batch_unpack: null compact table %d", k);
    t.dst[k] = d_compact[k];
  }
  for (int k = 0; k < n_grads; ++k) {
    SRH_REQUIRE(d_compact_grads[k], "batch_unpack: null gradient table %d", k);
    t.dst[n_tables + k] = d_compact_grads[k];
  }
  const int64_t total = (int64_t)kSegs * l.B * (dl / 4) * world * (n_tables + n_grads);
  unpack_kernel<<<(unsigned)((total + 255) / 256), 256, 0, srh::as_stream(stream)>>>(
      l, t, n_tables, n_grads, world, dl / 4, reinterpret_cast<const float4*>(d_recv));
  SRH_LAUNCH_CHECK();
  return SRH_OK;
}

srh_status_t srh_batch_scatter(const srh_batch_lists_t* lists, int32_t n_pairs, const float* const* d_compact_grads,
                               float* const* d_local_grads, int32_t d_full, int32_t col0, int32_t dl, void* stream) {
  Lists l;
  if (srh_status_t st = take_lists(lists, l, "batch_scatter")) return st;
  SRH_REQUIRE(n_pairs >= 1 && n_pairs <= SRH_MAX_EXCHANGE && d_compact_grads && d_local_grads,
              "batch_scatter: 1..%d (compact, local) pairs", SRH_MAX_EXCHANGE);
  SRH_REQUIRE(dl > 0 && col0 >= 0 && col0 + dl <= d_full, "batch_scatter: slice [%d, %d) outside %d columns", col0,
              col0 + dl, d_full);
  Ptrs4 t{};
  for (int k = 0; k < n_pairs; ++k) {
    SRH_REQUIRE(d_compact_grads[k] && d_local_grads[k], "batch_scatter: null pair %d", k);
    t.src[k] = d_compact_grads[k];
    t.dst[k] = d_local_grads[k];
  }
  const int64_t total = (int64_t)kSegs * l.B * dl * n_pairs;
  scatter_kernel<<<(unsigned)((total + 255) / 256), 256, 0, srh::as_stream(stream)>>>(l, t, n_pairs, d_full, col0, dl);
  SRH_LAUNCH_CHECK();
  return SRH_OK;
}

}  // extern "C"
