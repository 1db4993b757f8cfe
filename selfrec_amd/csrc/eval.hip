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
// FILTER: instead of writing the 32x32 score tile, keep only the scores that reach the row's threshold --
// appended to the row's candidate list (see srh_score_mask_topk_filtered); C is not touched.  (Training
// items are dropped from the lists afterwards, by cand_topk_kernel: a binary search per score in this
// epilogue stalled the MFMA loop.)
struct FilterArgs {
  const float* thr;            // row r's threshold = thr[r * thr_stride]
  int thr_stride;
  int32_t* cnt;                // per row: candidates seen (may exceed cap: the row is then incomplete)
  int32_t* cand_id;
  float* cand_sc;
  int cap;
};
constexpr int kStageCap = 192;   // survivors a wave stages in LDS before it appends them to the rows' lists

// append the wave's staged survivors to their rows' candidate lists: each lane takes one entry, one
// vector atomic reserves the slots
__device__ __forceinline__ void filter_flush(const FilterArgs& f, int m0, const short* stage_row, const int* stage_col,
                                             const float* stage_sc, int staged, int lane) {
  for (int e = lane; e < staged; e += 64) {
    const int row = m0 + stage_row[e];
    const int slot = atomicAdd(f.cnt + row, 1);
    if (slot < f.cap) {
      f.cand_id[(size_t)row * f.cap + slot] = stage_col[e];
      f.cand_sc[(size_t)row * f.cap + slot] = stage_sc[e];
    }
  }
}

template <int D, int TILES_PER_WAVE, bool FILTER>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const float* __restrict__ A, const int32_t* __restrict__ a_rows,
                                                      const float* __restrict__ B, float* __restrict__ C,
                                                      int m, int n, FilterArgs f) {
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
  __shared__ short s_stage_row[FILTER ? 4 * kStageCap : 1];
  __shared__ int s_stage_col[FILTER ? 4 * kStageCap : 1];
  __shared__ float s_stage_sc[FILTER ? 4 * kStageCap : 1];
  short* stage_row = s_stage_row + (FILTER ? wv * kStageCap : 0);
  int* stage_col = s_stage_col + (FILTER ? wv * kStageCap : 0);
  float* stage_sc = s_stage_sc + (FILTER ? wv * kStageCap : 0);
  int staged = 0;                // wave-uniform
  float thr[16];                 // FILTER: thresholds of the 16 rows this lane holds results for
  if (FILTER) {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int row = m0 + (t & 3) + 8 * (t >> 2) + 4 * h;
      thr[t] = (row < m) ? f.thr[(size_t)row * f.thr_stride] : INFINITY;
    }
  }
  // the B tile of step tt+1 is in flight while the 32 MFMAs of step tt run (one wave cannot hide an L2 /
  // Infinity-Cache round trip per tile behind anything else)
  float bn[DH];
  auto load_b = [&](int tile, float* dst) {
    const int br = min(tile * 32 + r32, n - 1);
    const float4* bp = reinterpret_cast<const float4*>(B + (size_t)br * D + h * DH);
#pragma unroll
    for (int t = 0; t < DH / 4; ++t) {
      float4 v = bp[t];
      dst[4 * t] = v.x; dst[4 * t + 1] = v.y; dst[4 * t + 2] = v.z; dst[4 * t + 3] = v.w;
    }
  };
  load_b(tile0, bn);
  for (int tt = 0; tt < TILES_PER_WAVE; ++tt) {
    const int tile = tile0 + tt;
    if (tile >= n_tiles) break;
    const int n0 = tile * 32;
    float b[DH];
#pragma unroll
    for (int t = 0; t < DH; ++t) b[t] = bn[t];
    if (tt + 1 < TILES_PER_WAVE && tile + 1 < n_tiles) load_b(tile + 1, bn);
    floatx16 acc;
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = 0.f;
#pragma unroll
    for (int s = 0; s < DH; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
    // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int col = n0 + r32;
    if (FILTER) {
      // survivors of this tile go to the wave's LDS staging list (ballot prefix: no atomics); the list is
      // flushed -- one vector atomic for up to 64 survivors -- only when it could overflow or at the end
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const bool pass = (col < n) && (acc[t] >= thr[t]);             // (rows >= m carry +inf)
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(pass);
        if (bal == 0) continue;                                         // wave-uniform
        if (pass) {
          const int at = staged + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0));
          stage_row[at] = (short)((t & 3) + 8 * (t >> 2) + 4 * h);
          stage_col[at] = col;
          stage_sc[at] = acc[t];
        }
        staged += __builtin_popcountll(bal);
        if (staged > kStageCap - 64) { filter_flush(f, m0, stage_row, stage_col, stage_sc, staged, lane); staged = 0; }
      }
    } else {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int row = m0 + (t & 3) + 8 * (t >> 2) + 4 * h;
        if (row < m && col < n) C[(size_t)row * n + col] = acc[t];
      }
    }
  }
  if (FILTER && staged > 0) filter_flush(f, m0, stage_row, stage_col, stage_sc, staged, lane);
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
  for (int p = s + lane; p < e; p += 64) {
    const int item = indices[p];
    if (item < n) scores[(size_t)q * n + item] = -10e8f;      // (n < catalogue size: a leading slice of the items)
  }
}

// Exact top-K of each row's candidate list (score desc, id asc) -- the ranking stage of topk_kernel on a
// few hundred survivors instead of the whole catalogue.  Rows whose list overflowed are left alone.
__global__ __launch_bounds__(256) void cand_topk_kernel(const int32_t* __restrict__ cnt, const int32_t* __restrict__ cand_id,
                                                        const float* __restrict__ cand_sc, int cap, int k,
                                                        const int32_t* __restrict__ user_ids, int user_base,
                                                        const int32_t* __restrict__ r_indptr,
                                                        const int32_t* __restrict__ r_indices,
                                                        int32_t* __restrict__ out_ids, float* __restrict__ out_scores) {
  extern __shared__ unsigned char cand_smem[];
  float* s_val = reinterpret_cast<float*>(cand_smem);
  int* s_idx = reinterpret_cast<int*>(cand_smem) + cap;
  __shared__ int s_n;
  const int row = blockIdx.x;
  const int c = cnt[row];
  if (c > cap) return;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  // training items of the user are dropped here (graph_recommender.py:49-50 masks them to -10e8)
  const int u = user_ids ? user_ids[row] : user_base + row;
  const int rs = r_indptr ? r_indptr[u] : 0, re = r_indptr ? r_indptr[u + 1] : 0;
  for (int t = threadIdx.x; t < c; t += 256) {
    const int id = cand_id[(size_t)row * cap + t];
    int lo = rs, hi = re;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (r_indices[mid] < id) lo = mid + 1; else hi = mid;
    }
    if (lo < re && r_indices[lo] == id) continue;
    const int at = atomicAdd(&s_n, 1);
    s_val[at] = cand_sc[(size_t)row * cap + t];
    s_idx[at] = id;
  }
  __syncthreads();
  const int nv = s_n;
  for (int t = threadIdx.x; t < nv; t += 256) {
    const float v = s_val[t];
    const int id = s_idx[t];
    int rank = 0;
    for (int j = 0; j < nv; ++j) rank += (s_val[j] > v) || (s_val[j] == v && s_idx[j] < id);
    if (rank < k) {
      out_ids[(size_t)row * k + rank] = id;
      out_scores[(size_t)row * k + rank] = v;
    }
  }
}


// ---------------------------------------------------------------------------------------
// The FILTER pass on the bf16 pipe (round 3).  The filter only has to decide "can this score reach the row's
// bound?" -- it never has to report a score -- so it runs on split-bf16 operands (x = hi + lo, three MFMAs of
// v_mfma_f32_32x32x16_bf16 per 16 dimensions: 5.3x the f32 MFMA's rate) against a bound lowered by a rigorous margin;
// the few hundred survivors per user are then RE-SCORED with the very instruction sequence of gemm_nt_kernel (so
// ids and scores stay bit-identical to the plain pipeline) and ranked.
//   |s~ - s| <= 3 * 2^-18 * sum_k |u_k i_k| (dropped lo.lo, two operand representations) + f32 accumulation of either
//   side (d * 2^-24 * sum |u_k i_k|) <= 4e-5 * |u| * max_i |i| for d <= 128 =: delta_u  (kFilterMargin)
// A true top-K item has s >= t_u, hence s~ >= t_u - delta_u: it survives.
// ---------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr float kFilterMargin = 4e-5f;

// rows of an f32 table as bf16 hi / lo images (row-major) + the row's L2 norm; the table's largest norm in *max_norm
// (float bits compare like unsigned ints for non-negative floats)
template <int LPR>
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ X, const int32_t* __restrict__ rows, int n,
                                                         uint16_t* __restrict__ hi, uint16_t* __restrict__ lo,
                                                         float* __restrict__ norm, unsigned int* __restrict__ max_norm) {
  constexpr int G = 64 / LPR;
  const int lane = threadIdx.x & 63, g = lane / LPR, sub = lane % LPR;
  const int r = (int)((blockIdx.x * 256u + threadIdx.x) >> 6) * G + g;
  const bool valid = r < n;
  const int src = valid ? (rows ? rows[r] : r) : 0;
  const float4 v = reinterpret_cast<const float4*>(X)[(size_t)src * LPR + sub];
  const float ss = group_sum<LPR>(f4_dot(v, v));
  if (!valid) return;
  const float f[4] = {v.x, v.y, v.z, v.w};
  uint16_t h[4], l[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const __bf16 bh = (__bf16)f[t];
    h[t] = __builtin_bit_cast(uint16_t, bh);
    l[t] = __builtin_bit_cast(uint16_t, (__bf16)(f[t] - (float)bh));
  }
  const size_t at = ((size_t)r * LPR + sub) * 4;
  *reinterpret_cast<uint2*>(hi + at) = make_uint2(h[0] | ((uint32_t)h[1] << 16), h[2] | ((uint32_t)h[3] << 16));
  *reinterpret_cast<uint2*>(lo + at) = make_uint2(l[0] | ((uint32_t)l[1] << 16), l[2] | ((uint32_t)l[3] << 16));
  if (sub == 0) {
    const float nr = sqrtf(ss);
    if (norm) norm[r] = nr;
    if (max_norm) atomicMax(max_norm, __float_as_uint(nr));
  }
}

struct Filter16Args {
  const float* thr;            // row r's exact bound = thr[r * thr_stride]
  int thr_stride;
  const float* u_norm;         // |u_r|
  const unsigned int* max_item_norm;
  int32_t* cnt;
  int32_t* cand_id;
  int cap;
};

// One wave: 64 query rows (two 32-row MFMA blocks, operands resident) x a stream of 32-item tiles.  Operand layout of
// v_mfma_f32_32x32x16_bf16: lane l gives A[i = l & 31][k = 8 (l >> 5) .. +7] and B[k = 8 (l >> 5) .. +7][j = l & 31]; k is
// a free summation index, so MFMA step s takes dimensions [16 s + 8 h, 16 s + 8 h + 8) from lane half h -- one 16-byte
// load per operand, image and step.  Two row blocks per B tile halve the bytes per MFMA: with one, four SIMDs of a CU
// would ask the vector L1 for 85 B/clk (it delivers 64).
template <int D, int TILES_PER_WAVE>
__global__ __launch_bounds__(256) void filter16_kernel(const uint16_t* __restrict__ Uhi, const uint16_t* __restrict__ Ulo,
                                                       const uint16_t* __restrict__ Ihi, const uint16_t* __restrict__ Ilo,
                                                       int m, int n, Filter16Args f) {
  constexpr int KS = D / 16;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r32 = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * 64;
  const int tile0 = (blockIdx.x * 4 + wv) * TILES_PER_WAVE;
  const int n_tiles = (n + 31) / 32;
  if (tile0 >= n_tiles) return;
  auto ld8 = [](const uint16_t* p) { return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(p)); };
  bf16x8 ah[2][KS], al[2][KS];
#pragma unroll
  for (int ub = 0; ub < 2; ++ub) {
    const int ar = min(m0 + 32 * ub + r32, m - 1);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      ah[ub][s] = ld8(Uhi + (size_t)ar * D + 16 * s + 8 * h);
      al[ub][s] = ld8(Ulo + (size_t)ar * D + 16 * s + 8 * h);
    }
  }
  __shared__ short s_stage_row[4 * kStageCap];
  __shared__ int s_stage_col[4 * kStageCap];
  short* stage_row = s_stage_row + wv * kStageCap;
  int* stage_col = s_stage_col + wv * kStageCap;
  int staged = 0;                // wave-uniform
  float thr[2][16];              // lowered bounds of the rows this lane holds results for
  {
    const float item_norm = __uint_as_float(*f.max_item_norm);
#pragma unroll
    for (int ub = 0; ub < 2; ++ub)
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int row = m0 + 32 * ub + (t & 3) + 8 * (t >> 2) + 4 * h;
        thr[ub][t] = (row < m) ? f.thr[(size_t)row * f.thr_stride] - kFilterMargin * f.u_norm[row] * item_norm : INFINITY;
      }
  }
  auto flush = [&]() {
    for (int e = lane; e < staged; e += 64) {
      const int row = m0 + stage_row[e];
      const int slot = atomicAdd(f.cnt + row, 1);
      if (slot < f.cap) f.cand_id[(size_t)row * f.cap + slot] = stage_col[e];
    }
    staged = 0;
  };
  bf16x8 bhn[KS], bln[KS];
  auto load_b = [&](int tile) {
    const int br = min(tile * 32 + r32, n - 1);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      bhn[s] = ld8(Ihi + (size_t)br * D + 16 * s + 8 * h);
      bln[s] = ld8(Ilo + (size_t)br * D + 16 * s + 8 * h);
    }
  };
  load_b(tile0);
  for (int tt = 0; tt < TILES_PER_WAVE; ++tt) {
    const int tile = tile0 + tt;
    if (tile >= n_tiles) break;
    bf16x8 bh[KS], bl[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) { bh[s] = bhn[s]; bl[s] = bln[s]; }
    if (tt + 1 < TILES_PER_WAVE && tile + 1 < n_tiles) load_b(tile + 1);      // in flight under this tile's MFMAs
    const int col = tile * 32 + r32;
#pragma unroll
    for (int ub = 0; ub < 2; ++ub) {
      floatx16 acc;
#pragma unroll
      for (int t = 0; t < 16; ++t) acc[t] = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ub][s], bh[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ub][s], bl[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ub][s], bh[s], acc, 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const bool pass = (col < n) && (acc[t] >= thr[ub][t]);            // (rows >= m carry +inf)
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(pass);
        if (bal == 0) continue;                                            // wave-uniform
        if (pass) {
          const int at = staged + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0));
          stage_row[at] = (short)(32 * ub + (t & 3) + 8 * (t >> 2) + 4 * h);
          stage_col[at] = col;
        }
        staged += __builtin_popcountll(bal);
        if (staged > kStageCap - 64) flush();
      }
    }
  }
  if (staged > 0) flush();
}

// Survivors of filter16_kernel: exact scores by the instruction sequence of gemm_nt_kernel (the user row replicated over
// the 32 A rows of the tile, 32 candidates as the B columns: every output row is the chain gemm_nt_kernel runs for that
// (user, item) -- same operands per MFMA, same order, same zero start, hence the same bits), training items dropped,
// exact (score desc, id asc) order.  One workgroup per query row; rows whose list overflowed are left alone.
template <int D>
__global__ __launch_bounds__(256) void rescore_topk_kernel(const float* __restrict__ U, const int32_t* __restrict__ user_ids,
                                                           int user_base, const float* __restrict__ I,
                                                           const int32_t* __restrict__ cnt, const int32_t* __restrict__ cand_id,
                                                           int cap, int k, const int32_t* __restrict__ r_indptr,
                                                           const int32_t* __restrict__ r_indices,
                                                           int32_t* __restrict__ out_ids, float* __restrict__ out_scores) {
  constexpr int DH = D / 2;
  extern __shared__ unsigned char cand_smem[];
  float* s_val = reinterpret_cast<float*>(cand_smem);
  int* s_idx = reinterpret_cast<int*>(cand_smem) + cap;
  __shared__ int s_n;
  const int row = blockIdx.x;
  const int c = cnt[row];
  if (c > cap) return;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r32 = lane & 31, h = lane >> 5;
  // U: the table user_ids index into, or the chunk's own rows (the caller passes the chunk base)
  const int u = user_ids ? user_ids[row] : user_base + row;          // the user's id: mask CSR row
  const int urow = user_ids ? u : row;
  float a[DH];
  {
    const float4* ap = reinterpret_cast<const float4*>(U + (size_t)urow * D + h * DH);
#pragma unroll
    for (int t = 0; t < DH / 4; ++t) {
      const float4 v = ap[t];
      a[4 * t] = v.x; a[4 * t + 1] = v.y; a[4 * t + 2] = v.z; a[4 * t + 3] = v.w;
    }
  }
  const int rs = r_indptr ? r_indptr[u] : 0, re = r_indptr ? r_indptr[u + 1] : 0;
  for (int t0 = wv * 32; t0 < c; t0 += 128) {
    const int ci = t0 + r32;
    const int id = cand_id[(size_t)row * cap + min(ci, c - 1)];
    float b[DH];
    {
      const float4* bp = reinterpret_cast<const float4*>(I + (size_t)id * D + h * DH);
#pragma unroll
      for (int t = 0; t < DH / 4; ++t) {
        const float4 v = bp[t];
        b[4 * t] = v.x; b[4 * t + 1] = v.y; b[4 * t + 2] = v.z; b[4 * t + 3] = v.w;
      }
    }
    floatx16 acc;
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = 0.f;
#pragma unroll
    for (int s = 0; s < DH; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
    if (h == 0 && ci < c) {
      // training items of the user are dropped here (graph_recommender.py:49-50 masks them to -10e8)
      int lo = rs, hi = re;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (r_indices[mid] < id) lo = mid + 1; else hi = mid;
      }
      if (!(lo < re && r_indices[lo] == id)) {
        const int at = atomicAdd(&s_n, 1);
        s_val[at] = acc[0];                         // (every row of the tile holds this user's score of column r32)
        s_idx[at] = id;
      }
    }
  }
  __syncthreads();
  const int nv = s_n;
  for (int t = threadIdx.x; t < nv; t += 256) {
    const float v = s_val[t];
    const int id = s_idx[t];
    int rank = 0;
    for (int j = 0; j < nv; ++j) rank += (s_val[j] > v) || (s_val[j] == v && s_idx[j] < id);
    if (rank < k) {
      out_ids[(size_t)row * k + rank] = id;
      out_scores[(size_t)row * k + rank] = v;
    }
  }
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

  // 16-byte loads over the aligned body of the row (any partition of the row into 256 disjoint sets
  // keeps the bound below valid), scalar tail
  const int n4 = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) ? n / 4 : 0;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  float mx = -INFINITY;
  for (int i = tid; i < n4; i += kTopkThreads) {
    const float4 v = x4[i];
    mx = fmaxf(fmaxf(mx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
  }
  for (int i = 4 * n4 + tid; i < n; i += kTopkThreads) mx = fmaxf(mx, x[i]);
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
  auto keep = [&](float v, int i) {
    if (v >= thr) {
      const int slot = atomicAdd(&s_cnt, 1);
      if (slot < kTopkCap) { s_val[slot] = v; s_idx[slot] = i; }
    }
  };
  for (int i = tid; i < n4; i += kTopkThreads) {
    const float4 v = x4[i];
    if (fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)) >= thr) {
      keep(v.x, 4 * i); keep(v.y, 4 * i + 1); keep(v.z, 4 * i + 2); keep(v.w, 4 * i + 3);
    }
  }
  for (int i = 4 * n4 + tid; i < n; i += kTopkThreads) keep(x[i], i);
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
srh_status_t launch_gemm(const float* a, const int32_t* a_rows, const float* b, float* c, int m, int n, hipStream_t st,
                         const FilterArgs* filter) {
  constexpr int TPW = 8;
  const int n_tiles = (n + 31) / 32;
  dim3 grid((n_tiles + 4 * TPW - 1) / (4 * TPW), (m + 31) / 32);
  if (filter) gemm_nt_kernel<D, TPW, true><<<grid, 256, 0, st>>>(a, a_rows, b, c, m, n, *filter);
  else gemm_nt_kernel<D, TPW, false><<<grid, 256, 0, st>>>(a, a_rows, b, c, m, n, FilterArgs{});
  SRH_LAUNCH_CHECK();
  return SRH_OK;
}

srh_status_t gemm_dispatch(const float* a, const int32_t* a_rows, const float* b, float* c, int64_t m, int64_t n,
                           int d, hipStream_t st, const FilterArgs* filter = nullptr) {
  SRH_REQUIRE(m > 0 && n > 0 && m < (int64_t(1) << 31) && n < (int64_t(1) << 31), "gemm_nt: bad shape");
  SRH_REQUIRE((m + 31) / 32 <= 65535, "gemm_nt: at most 2,097,120 query rows per call");
  switch (d) {
    case 32: return launch_gemm<32>(a, a_rows, b, c, (int)m, (int)n, st, filter);
    case 64: return launch_gemm<64>(a, a_rows, b, c, (int)m, (int)n, st, filter);
    case 128: return launch_gemm<128>(a, a_rows, b, c, (int)m, (int)n, st, filter);
    case 256: return launch_gemm<256>(a, a_rows, b, c, (int)m, (int)n, st, filter);
    default:
      srh::set_error("gemm_nt: d=%d unsupported (need 32, 64, 128 or 256)", d);
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

// Layout of one user chunk in the workspace of the filtered ranking
static inline int64_t filt_align(int64_t b) { return (b + 255) / 256 * 256; }
static int64_t filt_chunk_bytes(int64_t rows, int64_t sample, int32_t k, int32_t cap) {
  return filt_align(rows * sample * 4) + 2 * filt_align(rows * k * 4) + filt_align(rows * 4) + 2 * filt_align(rows * (int64_t)cap * 4);
}
// the split-bf16 filter's share: hi / lo images of the item table and of one chunk of query rows, their norms
static bool filt_split_served(int32_t d) { return d == 64 || d == 128; }
static int64_t filt_split_bytes(int64_t rows, int64_t n_items, int32_t d) {
  if (!filt_split_served(d)) return 0;
  return 2 * filt_align(n_items * d * 2) + 2 * filt_align(rows * d * 2) + filt_align(rows * 4) + 256;
}

int64_t srh_score_mask_topk_filtered_ws_bytes(int64_t chunk_rows, int64_t sample_items, int32_t k, int32_t cap,
                                              int64_t n_items, int32_t d) {
  if (chunk_rows <= 0 || sample_items <= 0 || k <= 0 || cap <= 0 || n_items <= 0 || d <= 0) return 0;
  return filt_chunk_bytes(chunk_rows, sample_items, k, cap) + filt_split_bytes(chunk_rows, n_items, d);
}

srh_status_t srh_score_mask_topk_filtered(const float* d_user_emb, const int32_t* d_user_ids, int64_t n_query,
                                          const float* d_item_emb, int64_t n_items, int32_t d,
                                          const int32_t* d_r_indptr, const int32_t* d_r_indices, int32_t k,
                                          int64_t sample_items, int32_t cap, int64_t chunk_rows, void* d_ws,
                                          int32_t* d_out_ids, float* d_out_scores, int32_t* d_out_counts, void* stream) {
  SRH_REQUIRE(d_user_emb && d_item_emb && d_ws && d_out_ids && d_out_scores && d_out_counts, "score_mask_topk_filtered: null argument");
  SRH_REQUIRE((d_r_indptr == nullptr) == (d_r_indices == nullptr), "score_mask_topk_filtered: mask CSR must be given whole");
  SRH_REQUIRE(n_query > 0 && n_items > 0 && chunk_rows > 0, "score_mask_topk_filtered: bad shape");
  SRH_REQUIRE(k >= 1 && k <= 128 && sample_items >= k && sample_items <= n_items && cap >= k && cap <= 4096,
              "score_mask_topk_filtered: need k <= sample_items <= n_items and k <= cap <= 4096");
  hipStream_t st = srh::as_stream(stream);
  char* ws = reinterpret_cast<char*>(d_ws);
  float* slab = reinterpret_cast<float*>(ws); ws += filt_align(chunk_rows * sample_items * 4);
  int32_t* s_ids = reinterpret_cast<int32_t*>(ws); ws += filt_align(chunk_rows * k * 4);
  float* s_sc = reinterpret_cast<float*>(ws); ws += filt_align(chunk_rows * k * 4);
  int32_t* cand_id = reinterpret_cast<int32_t*>(ws); ws += filt_align(chunk_rows * (int64_t)cap * 4);
  float* cand_sc = reinterpret_cast<float*>(ws); ws += filt_align(chunk_rows * (int64_t)cap * 4);
  // split-bf16 filter (d = 64 / 128): operand images of the whole item table once, of each chunk's query rows per chunk
  const bool split = filt_split_served(d);
  uint16_t *i_hi = nullptr, *i_lo = nullptr, *u_hi = nullptr, *u_lo = nullptr;
  float* u_norm = nullptr;
  unsigned int* max_norm = nullptr;
  if (split) {
    ws += filt_align(chunk_rows * 4);            // (the chunk layout's counter slot: counts live in d_out_counts)
    i_hi = reinterpret_cast<uint16_t*>(ws); ws += filt_align(n_items * d * 2);
    i_lo = reinterpret_cast<uint16_t*>(ws); ws += filt_align(n_items * d * 2);
    u_hi = reinterpret_cast<uint16_t*>(ws); ws += filt_align(chunk_rows * d * 2);
    u_lo = reinterpret_cast<uint16_t*>(ws); ws += filt_align(chunk_rows * d * 2);
    u_norm = reinterpret_cast<float*>(ws); ws += filt_align(chunk_rows * 4);
    max_norm = reinterpret_cast<unsigned int*>(ws);
    hipError_t err = hipMemsetAsync(max_norm, 0, sizeof(unsigned int), st);
    if (err != hipSuccess) { srh::set_error("score_mask_topk_filtered: %s", hipGetErrorString(err)); return SRH_ERR_HIP; }
    const int lpr = d / 4, g = 64 / lpr;
    const int blocks = (int)(((n_items + g - 1) / g + 3) / 4);
    if (d == 64) split_rows_kernel<16><<<blocks, 256, 0, st>>>(d_item_emb, nullptr, (int)n_items, i_hi, i_lo, nullptr, max_norm);
    else split_rows_kernel<32><<<blocks, 256, 0, st>>>(d_item_emb, nullptr, (int)n_items, i_hi, i_lo, nullptr, max_norm);
    SRH_LAUNCH_CHECK();
  }
  for (int64_t lo = 0; lo < n_query; lo += chunk_rows) {
    const int64_t m = std::min(chunk_rows, n_query - lo);
    const float* emb = d_user_ids ? d_user_emb : d_user_emb + lo * d;
    const int32_t* ids = d_user_ids ? d_user_ids + lo : nullptr;
    int32_t* cnt = d_out_counts + lo;
    // 1. exact top-K over a leading slice of the catalogue: its K-th score is a lower bound of the
    //    row's overall K-th score (a subset's K-th best cannot beat the whole set's)
    srh_status_t rc = gemm_dispatch(emb, ids, d_item_emb, slab, m, sample_items, d, st);
    if (rc) return rc;
    if (d_r_indptr) {
      mask_kernel<<<(int)((m + 3) / 4), 256, 0, st>>>(ids, (int)m, d_r_indptr, d_r_indices, slab, (int)sample_items, (int)lo);
      SRH_LAUNCH_CHECK();
    }
    rc = srh_topk_rows(slab, m, sample_items, k, s_ids, s_sc, stream);
    if (rc) return rc;
    // 2. all scores again, never stored: only those reaching the bound and not masked are kept
    hipError_t err = hipMemsetAsync(cnt, 0, sizeof(int32_t) * m, st);
    if (err != hipSuccess) { srh::set_error("score_mask_topk_filtered: %s", hipGetErrorString(err)); return SRH_ERR_HIP; }
    if (split) {
      // 2'. the filter on split-bf16 operands against the bound lowered by its error margin (ids only) ...
      const int lpr = d / 4, g = 64 / lpr;
      const int sb = (int)(((m + g - 1) / g + 3) / 4);
      constexpr int TPW = 8;
      const int n_tiles = (int)((n_items + 31) / 32);
      dim3 grid((n_tiles + 4 * TPW - 1) / (4 * TPW), (unsigned)((m + 63) / 64));
      Filter16Args f16{s_sc + (k - 1), k, u_norm, max_norm, cnt, cand_id, cap};
      if (d == 64) {
        split_rows_kernel<16><<<sb, 256, 0, st>>>(emb, ids, (int)m, u_hi, u_lo, u_norm, nullptr);
        filter16_kernel<64, TPW><<<grid, 256, 0, st>>>(u_hi, u_lo, i_hi, i_lo, (int)m, (int)n_items, f16);
      } else {
        split_rows_kernel<32><<<sb, 256, 0, st>>>(emb, ids, (int)m, u_hi, u_lo, u_norm, nullptr);
        filter16_kernel<128, TPW><<<grid, 256, 0, st>>>(u_hi, u_lo, i_hi, i_lo, (int)m, (int)n_items, f16);
      }
      SRH_LAUNCH_CHECK();
      // 3'. ... and the survivors re-scored by gemm_nt_kernel's own instruction sequence, masked, ranked
      if (d == 64)
        rescore_topk_kernel<64><<<(int)m, 256, (size_t)cap * 8, st>>>(emb, ids, (int)lo, d_item_emb, cnt, cand_id, cap, k,
                                                                      d_r_indptr, d_r_indices, d_out_ids + lo * k, d_out_scores + lo * k);
      else
        rescore_topk_kernel<128><<<(int)m, 256, (size_t)cap * 8, st>>>(emb, ids, (int)lo, d_item_emb, cnt, cand_id, cap, k,
                                                                       d_r_indptr, d_r_indices, d_out_ids + lo * k, d_out_scores + lo * k);
      SRH_LAUNCH_CHECK();
      continue;
    }
    FilterArgs fa{s_sc + (k - 1), k, cnt, cand_id, cand_sc, cap};
    rc = gemm_dispatch(emb, ids, d_item_emb, nullptr, m, n_items, d, st, &fa);
    if (rc) return rc;
    // 3. exact order of the survivors
    cand_topk_kernel<<<(int)m, 256, (size_t)cap * 8, st>>>(cnt, cand_id, cand_sc, cap, k, ids, (int)lo, d_r_indptr, d_r_indices,
                                                           d_out_ids + lo * k, d_out_scores + lo * k);
    SRH_LAUNCH_CHECK();
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
