// (a-2) Symmetric normalisation  A_hat = D^-1/2 A D^-1/2  on a device-resident CSR, for the
// full graph and for SGL's edge-dropped views.  Replaces reference data/graph.py:10-24
// (normalize_graph_mat, square branch) and data/ui_graph.py:58-65 (convert_to_laplacian_mat):
// a dropped view keeps the structure and zeroes the dropped entries, so no CSR rebuild or
// host round trip is needed (the reference rebuilds with scipy and re-uploads twice per
// epoch, SGL.py:28-29,89-96).
//
// HBM-bound integer/byte work: 2 passes over the structure, one wave per row.
// Algorithmic bytes: pass 1 nnz*(4 eid + 1 keep [+4 w]) + n*4 ; pass 2 nnz*(4 col + 4 eid + 1 keep
// [+4 w] + 4 val) + gathers of dinv (n*4, cache resident).
#include "common.h"

namespace {
using namespace srh;

__global__ __launch_bounds__(256) void degree_kernel(int n_rows, const int32_t* __restrict__ indptr,
                                                     const int32_t* __restrict__ edge_id,
                                                     const float* __restrict__ weight,
                                                     const uint8_t* __restrict__ keep,
                                                     const float* __restrict__ table, int table_len,
                                                     float* __restrict__ dinv_rows) {
  float* dinv = dinv_rows;       // (already offset to this shard's first row)
  const int row = (int)((blockIdx.x * 256u + threadIdx.x) >> 6);
  if (row >= n_rows) return;
  const int lane = threadIdx.x & 63;
  const int s = indptr[row], e = indptr[row + 1];
  float acc = 0.f;
  for (int p = s + lane; p < e; p += 64) {
    const bool k = keep ? (keep[edge_id[p]] != 0) : true;
    const float w = weight ? weight[p] : 1.0f;
    acc += k ? w : 0.f;
  }
  acc = wave_sum_f(acc);  // weights are small integers: exact in fp32 in any order
  if (lane == 0) {
    // numpy: np.power(rowsum, -0.5) in fp32, inf -> 0 (graph.py:14-15)
    float r = (acc > 0.f) ? (float)(1.0 / sqrt((double)acc)) : 0.f;
    const int k = (int)acc;
    if (table && acc >= 0.f && acc < (float)table_len && (float)k == acc) r = table[k];   // host numpy's value
    dinv[row] = r;
  }
}

__global__ __launch_bounds__(256) void normalize_kernel(int n_rows, const int32_t* __restrict__ indptr,
                                                        const int32_t* __restrict__ indices,
                                                        const int32_t* __restrict__ edge_id,
                                                        const float* __restrict__ weight,
                                                        const uint8_t* __restrict__ keep,
                                                        const float* __restrict__ dinv, int64_t row_offset,
                                                        float* __restrict__ vals) {
  const int row = (int)((blockIdx.x * 256u + threadIdx.x) >> 6);
  if (row >= n_rows) return;
  const int lane = threadIdx.x & 63;
  const int s = indptr[row], e = indptr[row + 1];
  const float dr = dinv[row_offset + row];
  for (int p = s + lane; p < e; p += 64) {
    const bool k = keep ? (keep[edge_id[p]] != 0) : true;
    const float w = weight ? weight[p] : 1.0f;
    // scipy evaluates (D^-1/2 A) D^-1/2 left to right in fp32 (graph.py:17-18)
    vals[p] = k ? (dr * w) * dinv[indices[p]] : 0.f;
  }
}
}  // namespace

extern "C" srh_status_t srh_adj_sym_normalize(int64_t n_rows, const int32_t* d_indptr,
                                              const int32_t* d_indices, const int32_t* d_edge_id,
                                              const float* d_weight, const uint8_t* d_keep,
                                              const float* d_inv_sqrt_table, int32_t table_len,
                                              float* d_deg_ws, float* d_vals, int64_t row_offset, int32_t phase,
                                              void* stream) {
  SRH_REQUIRE(d_indptr && d_indices && d_deg_ws && d_vals, "adj_sym_normalize: null argument");
  SRH_REQUIRE(n_rows > 0 && n_rows < (int64_t(1) << 31), "adj_sym_normalize: bad n_rows");
  SRH_REQUIRE(!d_keep || d_edge_id, "adj_sym_normalize: a keep mask needs edge ids");
  SRH_REQUIRE(row_offset >= 0 && phase >= 0 && phase <= 2, "adj_sym_normalize: bad row_offset / phase");
  hipStream_t st = srh::as_stream(stream);
  const int blocks = (int)((n_rows + 3) / 4);
  if (phase != 2) {
    degree_kernel<<<blocks, 256, 0, st>>>((int)n_rows, d_indptr, d_edge_id, d_weight, d_keep,
                                          d_inv_sqrt_table, d_inv_sqrt_table ? table_len : 0, d_deg_ws + row_offset);
    SRH_LAUNCH_CHECK();
  }
  if (phase != 1) {
    normalize_kernel<<<blocks, 256, 0, st>>>((int)n_rows, d_indptr, d_indices, d_edge_id, d_weight, d_keep, d_deg_ws,
                                             row_offset, d_vals);
    SRH_LAUNCH_CHECK();
  }
  return SRH_OK;
}
