// (a-10, a-11) Full-catalogue scoring, training-item mask and top-K -- replaces the per-user
// python loop of reference base/graph_recommender.py:46-53 (predict: XSimGCL.py:57-60, mask
// value -10e8, util/algorithm.py:144-156 find_k_largest).
//
//   scores = U_q (m x d) . I^T (d x n)  on v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, each
//   output a k-ordered fmaf chain -- exact-f32 numerics, so ranking parity needs no tolerance
//   beyond summation order.  Roofline: MFMA fp32 (157 TFLOP/s dense), 2*m*n*d flops.
//   Output tile traffic (4 B per 2*d flops) keeps the kernel at the MFMA/HBM ridge for d=64,
//   which is why callers chunk the users so the score slab stays in the 256 MiB Infinity Cache.
#include "common.h"

namespace {
using namespace srh;

typedef float floatx16 __attribute__((ext_vector_type(16)));

// C[m0+i][n0+j] for a 32x32 tile per wave; each wave keeps its 32 A rows in registers and
// walks TILES_PER_WAVE column tiles.  Operand layout of v_mfma_f32_32x32x2_f32: lane l gives
// A[i=l&31][k=l>>5] and B[k=l>>5][j=l&31]; the k index is a free summation index, so lane
// half h takes dimensions [h*D/2, (h+1)*D/2) of its row -- every load is a contiguous float4.
template <int D, int TILES_PER_WAVE>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const float* __restrict__ A, const int32_t* __restrict__ a_rows,
                                                      const float* __restrict__ B, float* __restrict__ C,
                                                      int m, int n) {
  constexpr int DH = D / 2;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r32 = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * 32;
  const int tile0 = (blockIdx.x * 4 + wv) * TILES_PER_WAVE;
  const int n_tiles = (n + 31) / 32;
  if (tile0 >= n_tiles) return;

  float a[DH];
  {
    const int ar = min(m0 + r32, m - 1);
    const int src = a_rows ? a_rows[ar] : ar;
    const float4* ap = reinterpret_cast<const float4*>(A + (size_t)src * D + h * DH);
#pragma unroll
    for (int t = 0; t < DH / 4; ++t) {
      float4 v = ap[t];
      a[4 * t] = v.x; a[4 * t + 1] = v.y; a[4 * t + 2] = v.z; a[4 * t + 3] = v.w;
    }
  }
  for (int tt = 0; tt < TILES_PER_WAVE; ++tt) {
    const int tile = tile0 + tt;
    if (tile >= n_tiles) break;
    const int n0 = tile * 32;
    float b[DH];
    {
      const int br = min(n0 + r32, n - 1);
      const float4* bp = reinterpret_cast<const float4*>(B + (size_t)br * D + h * DH);
#pragma unroll
      for (int t = 0; t < DH / 4; ++t) {
        float4 v = bp[t];
        b[4 * t] = v.x; b[4 * t + 1] = v.y; b[4 * t + 2] = v.z; b[4 * t + 3] = v.w;
      }
    }
    floatx16 acc;
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = 0.f;
#pragma unroll
    for (int s = 0; s < DH; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
    // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int col = n0 + r32;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int row = m0 + (t & 3) + 8 * (t >> 2) + 4 * h;
      if (row < m && col < n) C[(size_t)row * n + col] = acc[t];
    }
  }
}

// scores[q][item] = -1e9 for every training item of query user q (graph_recommender.py:49-50)
__global__ __launch_bounds__(256) void mask_kernel(const int32_t* __restrict__ user_ids, int m,
                                                   const int32_t* __restrict__ indptr,
                                                   const int32_t* __restrict__ indices, float* __restrict__ scores,
                                                   int n, int user_base) {
  const int q = (int)((blockIdx.x * 256u + threadIdx.x) >> 6);
  if (q >= m) return;
  const int lane = threadIdx.x & 63;
  const int u = user_ids ? user_ids[q] : user_base + q;
  const int s = indptr[u], e = indptr[u + 1];
  for (int p = s + lane; p < e; p += 64) scores[(size_t)q * n + indices[p]] = -10e8f;
}

// ---------------------------------------------------------------------------------------
// top-K of one row per workgroup.  Ordering: score descending, then id ascending.
//   pass A: every thread's running maximum; the K-th largest of those 256 maxima is a lower
//           bound t of the row's K-th largest element (they are 256 distinct elements).
//   pass B: elements >= t are appended to an LDS candidate list (a few dozen for real data).
//   rank  : each candidate counts the candidates that precede it; rank < K writes slot rank.
//   If the candidate list overflows (many ties), fall back to K rounds of block arg-max.
// ---------------------------------------------------------------------------------------
constexpr int kTopkThreads = 256;
constexpr int kTopkCap = 2048;

__device__ __forceinline__ bool before(float sa, int ia, float sb, int ib) {
  return (sa > sb) || (sa == sb && ia < ib);
}

__global__ __launch_bounds__(kTopkThreads) void topk_kernel(const float* __restrict__ scores, int rows, int n, int k,
                                                            int32_t* __restrict__ out_ids, float* __restrict__ out_scores) {
  __shared__ float s_val[kTopkCap];
  __shared__ int s_idx[kTopkCap];
  __shared__ float s_max[kTopkThreads];
  __shared__ int s_cnt;
  __shared__ float s_thr;
  __shared__ float s_bs[kTopkThreads / 64];
  __shared__ int s_bi[kTopkThreads / 64];
  __shared__ float s_last_s;
  __shared__ int s_last_i;

  const int row = blockIdx.x;
  if (row >= rows) return;
  const float* x = scores + (size_t)row * n;
  const int tid = threadIdx.x;

  float mx = -INFINITY;
  for (int i = tid; i < n; i += kTopkThreads) mx = fmaxf(mx, x[i]);
  s_max[tid] = mx;
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  {
    int rank = 0;
    for (int j = 0; j < kTopkThreads; ++j) {
      const float o = s_max[j];
      rank += (o > mx) || (o == mx && j < tid);
    }
    if (rank == min(k, kTopkThreads) - 1) s_thr = mx;
  }
  __syncthreads();
  const float thr = s_thr;
  for (int i = tid; i < n; i += kTopkThreads) {
    const float v = x[i];
    if (v >= thr) {
      const int slot = atomicAdd(&s_cnt, 1);
      if (slot < kTopkCap) { s_val[slot] = v; s_idx[slot] = i; }
    }
  }
  __syncthreads();
  const int cnt = s_cnt;
  if (cnt <= kTopkCap) {
    for (int c = tid; c < cnt; c += kTopkThreads) {
      const float v = s_val[c];
      const int id = s_idx[c];
      int rank = 0;
      for (int j = 0; j < cnt; ++j) rank += before(s_val[j], s_idx[j], v, id);
      if (rank < k) {
        out_ids[(size_t)row * k + rank] = id;
        out_scores[(size_t)row * k + rank] = v;
      }
    }
    return;
  }
  // tie-heavy row: K rounds of "largest element that comes after the previous pick"
  if (tid == 0) { s_last_s = INFINITY; s_last_i = -1; }
  __syncthreads();
  for (int r = 0; r < k; ++r) {
    const float ls = s_last_s;
    const int li = s_last_i;
    float bs = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = tid; i < n; i += kTopkThreads) {
      const float v = x[i];
      const bool after_last = before(ls, li, v, i);
      if (after_last && before(v, i, bs, bi)) { bs = v; bi = i; }
    }
    for (int msk = 1; msk < 64; msk <<= 1) {
      const float os = __shfl_xor(bs, msk);
      const int oi = __shfl_xor(bi, msk);
      if (before(os, oi, bs, bi)) { bs = os; bi = oi; }
    }
    if ((tid & 63) == 0) { s_bs[tid >> 6] = bs; s_bi[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
      for (int wv = 1; wv < kTopkThreads / 64; ++wv)
        if (before(s_bs[wv], s_bi[wv], bs, bi)) { bs = s_bs[wv]; bi = s_bi[wv]; }
      out_ids[(size_t)row * k + r] = bi;
      out_scores[(size_t)row * k + r] = bs;
      s_last_s = bs;
      s_last_i = bi;
    }
    __syncthreads();
  }
}

template <int D>
srh_status_t launch_gemm(const float* a, const int32_t* a_rows, const float* b, float* c, int m, int n, hipStream_t st) {
  constexpr int TPW = 8;
  const int n_tiles = (n + 31) / 32;
  dim3 grid((n_tiles + 4 * TPW - 1) / (4 * TPW), (m + 31) / 32);
  gemm_nt_kernel<D, TPW><<<grid, 256, 0, st>>>(a, a_rows, b, c, m, n);
  SRH_LAUNCH_CHECK();
  return SRH_OK;
}

srh_status_t gemm_dispatch(const float* a, const int32_t* a_rows, const float* b, float* c, int64_t m, int64_t n,
                           int d, hipStream_t st) {
  SRH_REQUIRE(m > 0 && n > 0 && m < (int64_t(1) << 31) && n < (int64_t(1) << 31), "gemm_nt: bad shape");
  SRH_REQUIRE((m + 31) / 32 <= 65535, "gemm_nt: at most 2,097,120 query rows per call");
  switch (d) {
    case 32: return launch_gemm<32>(a, a_rows, b, c, (int)m, (int)n, st);
    case 64: return launch_gemm<64>(a, a_rows, b, c, (int)m, (int)n, st);
    case 128: return launch_gemm<128>(a, a_rows, b, c, (int)m, (int)n, st);
    default:
      srh::set_error("gemm_nt: d=%d unsupported (need 32, 64 or 128)", d);
      return SRH_ERR_UNSUPPORTED;
  }
}


// (f-3) hit flags for the metric tail: flag[q][r] = 1 iff ranked id ids[q][r] is a test item of the
// query's user (binary search in that user's sorted test row) -- what Metric.hits / Metric.NDCG
// (reference util/evaluation.py:7-16,66-78) find by python set membership.
__global__ __launch_bounds__(256) void hit_flags_kernel(const int32_t* __restrict__ ids, int64_t total, int k,
                                                        const int32_t* __restrict__ user_ids,
                                                        const int32_t* __restrict__ t_indptr,
                                                        const int32_t* __restrict__ t_indices,
                                                        uint8_t* __restrict__ flags) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int64_t q = t / k;
  const int u = user_ids ? user_ids[q] : (int)q;
  const int item = ids[t];
  int lo = t_indptr[u], hi = t_indptr[u + 1];
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (t_indices[mid] < item) lo = mid + 1; else hi = mid;
  }
  flags[t] = (lo < t_indptr[u + 1] && t_indices[lo] == item) ? 1 : 0;
}
}  // namespace

extern "C" {

srh_status_t srh_gemm_nt_f32(const float* d_a, const float* d_b, float* d_c, int64_t m, int64_t n, int32_t d,
                             void* stream) {
  SRH_REQUIRE(d_a && d_b && d_c, "gemm_nt: null argument");
  return gemm_dispatch(d_a, nullptr, d_b, d_c, m, n, d, srh::as_stream(stream));
}

srh_status_t srh_topk_rows(const float* d_scores, int64_t rows, int64_t n, int32_t k, int32_t* d_out_ids,
                           float* d_out_scores, void* stream) {
  SRH_REQUIRE(d_scores && d_out_ids && d_out_scores, "topk_rows: null argument");
  SRH_REQUIRE(rows > 0 && rows < (int64_t(1) << 31) && n > 0 && n < (int64_t(1) << 31), "topk_rows: bad shape");
  SRH_REQUIRE(k >= 1 && k <= 128 && k <= n, "topk_rows: k=%d must be in [1, min(128, n)]", k);
  topk_kernel<<<(int)rows, kTopkThreads, 0, srh::as_stream(stream)>>>(d_scores, (int)rows, (int)n, k, d_out_ids, d_out_scores);
  SRH_LAUNCH_CHECK();
  return SRH_OK;
}

srh_status_t srh_score_mask_topk(const float* d_user_emb, const int32_t* d_user_ids, int64_t n_query,
                                 const float* d_item_emb, int64_t n_items, int32_t d,
                                 const int32_t* d_r_indptr, const int32_t* d_r_indices, int32_t k,
                                 float* d_scores_ws, int64_t ws_rows, int32_t* d_out_ids, float* d_out_scores,
                                 void* stream) {
  SRH_REQUIRE(d_user_emb && d_item_emb && d_scores_ws && d_out_ids && d_out_scores, "score_mask_topk: null argument");
  SRH_REQUIRE((d_r_indptr == nullptr) == (d_r_indices == nullptr), "score_mask_topk: mask CSR must be given whole");
  SRH_REQUIRE(ws_rows > 0, "score_mask_topk: the score slab must hold at least one row");
  hipStream_t st = srh::as_stream(stream);
  // the queries go through the slab ws_rows at a time (it is sized to stay in the Infinity Cache)
  for (int64_t lo = 0; lo < n_query; lo += ws_rows) {
    const int64_t m = std::min(ws_rows, n_query - lo);
    const float* emb = d_user_ids ? d_user_emb : d_user_emb + lo * d;
    const int32_t* ids = d_user_ids ? d_user_ids + lo : nullptr;
    srh_status_t rc = gemm_dispatch(emb, ids, d_item_emb, d_scores_ws, m, n_items, d, st);
    if (rc) return rc;
    if (d_r_indptr) {
      mask_kernel<<<(int)((m + 3) / 4), 256, 0, st>>>(ids, (int)m, d_r_indptr, d_r_indices, d_scores_ws, (int)n_items,
                                                      (int)lo);
      SRH_LAUNCH_CHECK();
    }
    rc = srh_topk_rows(d_scores_ws, m, n_items, k, d_out_ids + lo * k, d_out_scores + lo * k, stream);
    if (rc) return rc;
  }
  return SRH_OK;
}

srh_status_t srh_topk_hit_flags(const int32_t* d_ids, int64_t n_query, int32_t k, const int32_t* d_user_ids,
                                const int32_t* d_t_indptr, const int32_t* d_t_indices, uint8_t* d_flags,
                                void* stream) {
  SRH_REQUIRE(d_ids && d_t_indptr && d_t_indices && d_flags, "topk_hit_flags: null argument");
  SRH_REQUIRE(n_query > 0 && k >= 1 && n_query * k < (int64_t(1) << 40), "topk_hit_flags: bad shape");
  const int64_t total = n_query * k;
  hit_flags_kernel<<<(unsigned)((total + 255) / 256), 256, 0, srh::as_stream(stream)>>>(d_ids, total, k, d_user_ids,
                                                                                         d_t_indptr, d_t_indices, d_flags);
  SRH_LAUNCH_CHECK();
  return SRH_OK;
}

}  // extern "C"
