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
#include <hipcub/hipcub.hpp>

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
// The FILTER pass on the bf16 pipe (round 3; margins per item and one-term products: round 4).  The filter only has to
// decide "can this score reach the row's bound?" -- it never has to report a score -- so it runs on bf16 operands against
// rigorous error bounds; the few dozen survivors per user are then RE-SCORED with the very instruction sequence of
// gemm_nt_kernel (so ids and scores stay bit-identical to the plain pipeline) and ranked.
//   s~_uj = the bf16 product the MFMAs form,  |s~_uj - s_uj| <= delta_uj = c |u| |i_j|,   c = kFilterMargin:
//   SRH_F16_TERMS  products per 16 dimensions                 c
//   3              u_hi.i_hi + u_hi.i_lo + u_lo.i_hi           5e-5     (dropped lo.lo: 3 * 2^-18)
//   2              u_hi.i_hi + u_lo.i_hi                       1.99e-3  (the item side rounded to bf16: 2^-9 + 2^-18)
//   1 (default)    u_hi.i_hi                                   3.94e-3  (both sides rounded: 2^-8 + 2^-18)
//   -- |fl_bf16(x) - x| <= 2^-9 |x| (round to nearest even, 8 significant bits), the products are exact in f32, Cauchy-Schwarz
//   over the d terms, and the f32 accumulation of d + 1 <= 129 addends (the accumulator starts at -T, see below) adds at most
//   2 (d + 1) 2^-24 |u||i_j| = 1.6e-5 |u||i_j| wherever it matters (|T| <= |u||i_j| for any item that can reach T).
// The margin is PER ITEM: with one bound for the whole catalogue (round 3: c |u| max_j |i_j|) a few popular items of large
// norm -- trained tables have them: 100x the typical norm after 1300 steps -- widen everybody's margin, harmless at
// c = 4e-5 and ruinous at 4e-3 (984 of 31.5 k users overflowed their candidate lists).  Per item the margin scales with the
// score it guards.  The margin only decides how many items reach the exact re-score: near a row's K-th best score the
// catalogue is thin, so a 100x wider one adds a handful of survivors per user (50 -> 60) while the MFMA work and the item
// image staged through LDS drop to a third / a half.
//   bound stage:  slab[u][j] = s~_uj - delta_uj <= s_uj for a leading slice of the catalogue; T_u = a lower bound of its
//                 K-th largest entry (training items masked) <= the exact K-th best score of the row;
//   filter:       a top-K item has s_uj >= T_u, hence s~_uj + delta_uj >= T_u: the accumulator starts at -T_u and the test
//                 is fma(c |u|, |i_j|, acc) >= 0 -- one FMA per output and ONE comparison per 32 x 32 block (the maximum);
//   re-score:     L_j = s~ - delta, U_j = s~ + delta; tau = the K-th largest L_j of the unmasked survivors <= the exact K-th
//                 best; only survivors with U_j >= tau can be in the exact top-K: those are re-scored exactly and ranked.
// Where a ranking's 1.05 ms go (31.5 k users, trained tables, tools/gpu_session.sh evalbreakab with -DSRH_F16_EXP=1 / 2 builds):
// MFMAs + operand staging of the filter 0.17, its epilogue 0.25 (four blocks in five hold a survivor: 1.6 per 32 x 32 block),
// writing the survivors out 0.10, the re-score 0.19, the bound stage (slab, mask, bound) 0.21, the rest 0.13.  Tried on top
// and dropped, all within +-0.07 ms of this form: survivor lists private to (row, item range) with LDS slot counters instead
// of device-scope atomics (the atomics are not what the flush costs), long staging lists flushed when half full (seven
// waves wait at the stage barrier for the one that flushes), four returning atomics in flight per lane, wave-uniform
// ballots per row quad and output instead of the bit mask + select loop.
// ---------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#ifndef SRH_F16_TERMS
#define SRH_F16_TERMS 1
#endif
constexpr int kF16Terms = SRH_F16_TERMS;
static_assert(kF16Terms >= 1 && kF16Terms <= 3, "SRH_F16_TERMS: 1, 2 or 3");
constexpr float kFilterMargin = kF16Terms == 3 ? 5e-5f : (kF16Terms == 2 ? 1.99e-3f : 3.94e-3f);
// the survivor's score is handed on as acc + T (the accumulator held s~ - T): one more rounding of size 2^-24 (|T| + |s~|)
constexpr float kFilterAbsSlack = 3e-7f;

// The bound slice is the S items of LARGEST NORM, not the first S of the catalogue: a user's best scores sit on popular items,
// popular items are the ones training has pushed outwards, and the K-th best score over such a slice is a far tighter bound
// (trained XSimGCL tables, Yelp2018 shape, tools/sample_choice_probe.py: 2048 largest-norm items leave 27.7 items per user at or
// above the bound, the first 4096 of the catalogue 38.6, the first 2048 66.7).  So the item image is built in norm order --
// norms, a radix sort of (norm bits, item id) pairs, the permutation and its inverse: four small launches per ranking -- and
// everything downstream works on image rows; only the survivors' ids are translated back, when they are written out.
template <int LPR>
__global__ __launch_bounds__(256) void item_norms_kernel(const float* __restrict__ X, int n, uint32_t* __restrict__ norm_bits,
                                                         int32_t* __restrict__ iota) {
  constexpr int G = 64 / LPR;
  const int lane = threadIdx.x & 63, g = lane / LPR, sub = lane % LPR;
  const int r = (int)((blockIdx.x * 256u + threadIdx.x) >> 6) * G + g;
  const bool valid = r < n;
  const float4 v = reinterpret_cast<const float4*>(X)[(size_t)(valid ? r : 0) * LPR + sub];
  const float ss = group_sum<LPR>(f4_dot(v, v));
  if (valid && sub == 0) {
    norm_bits[r] = __float_as_uint(sqrtf(ss));                  // (non-negative floats order like their bit patterns)
    iota[r] = r;
  }
}
__global__ __launch_bounds__(256) void invert_order_kernel(const int32_t* __restrict__ order, int n, int32_t* __restrict__ column_of_item) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r < n) column_of_item[order[r]] = r;
}
static size_t filt_sort_temp_bytes(int64_t n) {
  size_t bytes = 0;
  (void)hipcub::DeviceRadixSort::SortPairsDescending(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const int32_t*)nullptr,
                                                     (int32_t*)nullptr, (int)n);
  return bytes;
}

// rows of an f32 table as bf16 hi / lo images + the row's L2 norm; the table's largest norm in *max_norm (float bits
// compare like unsigned ints for non-negative floats).  FRAG = false: row-major images (the query rows: each wave loads
// its A operands once).  FRAG = true: FRAGMENT-LINEAR images of the item table -- per 32-row tile, per image, per MFMA
// step s one contiguous 1 KB fragment whose lane l = 32 h + r holds row 32 tile + r, dimensions [16 s + 8 h, + 8): exactly
// what lane l feeds v_mfma_f32_32x32x16_bf16 as its B operand, so a workgroup copies whole tiles into LDS with direct
// global -> LDS loads and every wave reads its operand with one conflict-free ds_read_b128.
template <int LPR, bool FRAG>
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ X, const int32_t* __restrict__ rows, int n,
                                                         uint16_t* __restrict__ hi, uint16_t* __restrict__ lo,
                                                         float* __restrict__ norm, unsigned int* __restrict__ max_norm) {
  constexpr int G = 64 / LPR, D = 4 * LPR, KS = D / 16;
  const int lane = threadIdx.x & 63, g = lane / LPR, sub = lane % LPR;
  const int r = (int)((blockIdx.x * 256u + threadIdx.x) >> 6) * G + g;
  const bool valid = r < n;
  const int src = valid ? (rows ? rows[r] : r) : 0;
  const float4 v = reinterpret_cast<const float4*>(X)[(size_t)src * LPR + sub];
  const float ss = group_sum<LPR>(f4_dot(v, v));
  if (max_norm) {
    // one atomic per workgroup: same-address atomics serialise at ~11 ns each on this chip (one per row: 111 us for
    // 38 k items)
    __shared__ float s_mx[4];
    float mx = valid ? sqrtf(ss) : 0.f;
    for (int msk = 1; msk < 64; msk <<= 1) mx = fmaxf(mx, __shfl_xor(mx, msk));
    if (lane == 0) s_mx[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(max_norm, __float_as_uint(fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]))));
  }
  if (!valid) return;
  const float f[4] = {v.x, v.y, v.z, v.w};
  uint16_t h[4], l[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const __bf16 bh = (__bf16)f[t];
    h[t] = __builtin_bit_cast(uint16_t, bh);
    l[t] = __builtin_bit_cast(uint16_t, (__bf16)(f[t] - (float)bh));
  }
  const uint2 ph = make_uint2(h[0] | ((uint32_t)h[1] << 16), h[2] | ((uint32_t)h[3] << 16));
  const uint2 pl = make_uint2(l[0] | ((uint32_t)l[1] << 16), l[2] | ((uint32_t)l[3] << 16));
  if (FRAG) {
    const int col0 = 4 * sub, st = col0 >> 4, hh = (col0 >> 3) & 1, e0 = col0 & 7;
    const size_t tile = (size_t)(r >> 5);
    const size_t at_hi = ((tile * 2 + 0) * KS + st) * 512 + (size_t)(hh * 32 + (r & 31)) * 8 + e0;
    const size_t at_lo = ((tile * 2 + 1) * KS + st) * 512 + (size_t)(hh * 32 + (r & 31)) * 8 + e0;
    *reinterpret_cast<uint2*>(hi + at_hi) = ph;            // (hi: the base of the interleaved image; lo unused)
    *reinterpret_cast<uint2*>(hi + at_lo) = pl;
  } else {
    const size_t at = ((size_t)r * LPR + sub) * 4;
    *reinterpret_cast<uint2*>(hi + at) = ph;
    *reinterpret_cast<uint2*>(lo + at) = pl;
  }
  if (sub == 0) {
    const float nr = sqrtf(ss);
    if (norm) norm[r] = nr;
  }
}

struct Filter16Args {
  const float* thr;            // row r's exact bound = thr[r * thr_stride]
  int thr_stride;
  const float* u_norm;         // |u_r|
  const float* item_norm;      // |i_j| per IMAGE row, whole 32-item tiles (0 beyond the catalogue)
  const int32_t* order;        // image row -> item id (the image holds the items by norm, largest first)
  int32_t* cnt;
  int32_t* cand_id;
  float* cand_sc;              // the split-bf16 score of the candidate (within delta_u of the exact one)
  int cap;
};

__device__ __forceinline__ void glds16_b(const void* gsrc, void* ldst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)ldst, 16, 0, 0);
}
#ifndef SRH_F16_EXP
#define SRH_F16_EXP 0
#endif
constexpr int kF16StageCap = 256;   // survivors a wave collects in LDS between two flushes (one per stage of item tiles)
constexpr int kF16NormBytes = 1024;  // the staged tiles' item norms: <= 8 tiles x 32 floats per stage
constexpr int kF16Lds = 2 * (32768 + kF16NormBytes) + 4 * kF16StageCap * (4 + 4 + 2);   // two stages (+ norms) + the survivor lists
#ifndef SRH_F16_UB
#define SRH_F16_UB 1                // 32-row MFMA blocks per wave: 1 (8 waves per 256-row workgroup) or 2 (4 waves; the first version)
#endif

// One workgroup: 256 query rows (UB = 1: eight waves x one 32-row MFMA block, 110 VGPRs, four waves per SIMD with two workgroups
// per CU -- 400 -> 303 us per 16384-row chunk against UB = 2, the first version: four waves x two blocks, 189 VGPRs, two waves per
// SIMD, where one wave's operand reads / epilogue and the other's MFMAs overlapped little; operands resident in registers) against a
// contiguous range of 32-item tiles, which it copies into LDS a stage of ST tiles at a time (double-buffered: the copy
// of stage k + 1 runs under the MFMAs of stage k) -- one L2 read of the item images per 256 query rows instead of one
// per 64, and no wave ever waits for a global round trip per tile (the first version, every wave streaming its own B
// tiles from L2 with one tile of prefetch, ran at 217 us per 4096 x 38048 chunk for 24 us of MFMA work).
// Operand layout of v_mfma_f32_32x32x16_bf16: lane l gives A[i = l & 31][k = 8 (l >> 5) .. +7], B[k ..][j = l & 31]; k is a
// free summation index: MFMA step s takes dimensions [16 s + 8 h, + 8) from lane half h.
// SLAB: instead of filtering, write the split-bf16 scores of the item range into a (m x n) slab -- the bound stage (the
// K-th score over a leading slice of the catalogue), which then costs a quarter of the f32 GEMM it replaces; the bound
// derived from approximate scores is lowered by one more delta_u (see srh_score_mask_topk_filtered).
template <int D, bool SLAB = false, int UB = 2>
__global__ __launch_bounds__(512 / UB) void filter16_kernel(const uint16_t* __restrict__ Uhi, const uint16_t* __restrict__ Ulo,
                                                       const uint16_t* __restrict__ Ifrag, int m, int n,
                                                       Filter16Args f, float* __restrict__ C = nullptr) {
  constexpr int KS = D / 16;
  constexpr int TILE_BYTES = 2 * KS * 1024;          // hi fragments then lo fragments of one 32-item tile (the global image)
  constexpr int FR = kF16Terms == 3 ? 2 * KS : KS;   // fragments of a tile this build stages: hi + lo, or hi only
  // tiles per stage: 8 (d = 64 hi-only; 4 with the lo fragments), 4 (d = 128; 2).  The slab launch -- a slice of ~100 tiles dealt
  // round-robin over 8 workgroups per row block -- takes half-size stages so that the deal comes out even.
  constexpr int ST = (SLAB ? 16384 : 32768) / (FR * 1024);
  extern __shared__ __attribute__((aligned(16))) unsigned char f16_smem[];
  constexpr int STAGE_BYTES = ST * FR * 1024 + kF16NormBytes;      // 32 KB of fragments, then the tiles' item norms
  static_assert(ST * 128 <= kF16NormBytes, "norm slot of a stage");
  constexpr int WAVES = 8 / UB, CAP = kF16StageCap * UB / 2;       // (same LDS either way: 8 shorter lists or 4 longer ones)
  int* s_col = reinterpret_cast<int*>(f16_smem + 2 * STAGE_BYTES);
  float* s_sc = reinterpret_cast<float*>(s_col + WAVES * CAP);
  short* s_row = reinterpret_cast<short*>(s_sc + WAVES * CAP);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r32 = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * 256 + wv * 32 * UB;
  const int n_tiles = (n + 31) / 32;
  // The item image is in NORM order (largest first), and that is where a row's survivors are: the workgroups of one row block
  // take the STAGES of item tiles round-robin -- x, x + X, x + 2 X, ... -- so that each of them sees large and small norms
  // alike.  (Contiguous ranges gave the first workgroup of every row block nearly all of the survivor work, and with one wave
  // of 512 workgroups on the chip the slowest one IS the kernel's time.)
  const int t_end = n_tiles;
  const int t_begin = blockIdx.x * ST, t_step = gridDim.x * ST;
  if (t_begin >= t_end) return;
  const bool live = m0 < m;                          // (a wave beyond the chunk's rows still copies and synchronises)
  auto ld8 = [](const uint16_t* p) { return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(p)); };
  bf16x8 ah[UB][KS], al[UB][KS];
  float thr[UB][16];             // T_row of the rows this lane holds results for (+inf beyond the chunk: nothing passes)
  float cu[UB][16];              // c |u_row|: times the item's norm = the margin of one score
  {
#pragma unroll
    for (int ub = 0; ub < UB; ++ub) {
      const int ar = min(m0 + 32 * ub + r32, m - 1);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        ah[ub][s] = ld8(Uhi + (size_t)ar * D + 16 * s + 8 * h);
        if (kF16Terms >= 2) al[ub][s] = ld8(Ulo + (size_t)ar * D + 16 * s + 8 * h);
      }
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int row = m0 + 32 * ub + (t & 3) + 8 * (t >> 2) + 4 * h;
        cu[ub][t] = (row < m) ? kFilterMargin * f.u_norm[row] : 0.f;
        thr[ub][t] = SLAB ? 0.f : ((row < m) ? f.thr[(size_t)row * f.thr_stride] : INFINITY);
      }
    }
  }
  int* stage_col = s_col + wv * CAP;
  float* stage_sc = s_sc + wv * CAP;
  short* stage_row = s_row + wv * CAP;
  int staged = 0;                // wave-uniform
  auto flush = [&]() {
#if SRH_F16_EXP == 1                          // (timing experiment: what the survivors' global atomics + stores cost)
    staged = 0;
    return;
#endif
    for (int e = lane; e < staged; e += 64) {
      const int row = m0 + stage_row[e];
      const int slot = atomicAdd(f.cnt + row, 1);
      if (slot < f.cap) {
        f.cand_id[(size_t)row * f.cap + slot] = f.order[stage_col[e]];        // image row -> item id
        f.cand_sc[(size_t)row * f.cap + slot] = stage_sc[e];
      }
    }
    staged = 0;
  };
  const unsigned char* img = reinterpret_cast<const unsigned char*>(Ifrag);
  auto copy_stage = [&](int t0, unsigned char* dst) {           // tiles [t0, min(t0 + ST, t_end)) -> dst
    const int frags = min(ST, t_end - t0) * FR;
    const unsigned char* src = img + (size_t)t0 * TILE_BYTES;
    for (int k = wv; k < frags; k += WAVES) {
      // (hi-only builds skip the lo half of every tile of the global image)
      const size_t from = (FR == 2 * KS) ? (size_t)k * 1024 : (size_t)(k / KS) * TILE_BYTES + (size_t)(k % KS) * 1024;
      glds16_b(src + from + lane * 16, dst + k * 1024);
    }
    // the tiles' item norms (128 B per tile) behind the fragments: one 16-byte direct load per lane of one wave
    if (wv == (frags % WAVES) && lane * 4 < min(ST, t_end - t0) * 32)
      glds16_b(reinterpret_cast<const unsigned char*>(f.item_norm) + (size_t)t0 * 128 + lane * 16, dst + ST * FR * 1024);
  };
  copy_stage(t_begin, f16_smem);
  int cur = 0;
  for (int t0 = t_begin; t0 < t_end; t0 += t_step, cur ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's share of stage t0 (and its flush) landed
    __syncthreads();                                             // ... everyone's; and the other buffer has been read
    if (t0 + t_step < t_end) copy_stage(t0 + t_step, f16_smem + (cur ^ 1) * STAGE_BYTES);   // in flight under this stage's MFMAs
    if (live) {
      const unsigned char* buf = f16_smem + cur * STAGE_BYTES;
      const int nt = min(ST, t_end - t0);
      for (int tt = 0; tt < nt; ++tt) {
        bf16x8 bh[KS], bl[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          bh[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(buf + (tt * FR + s) * 1024 + lane * 16));
          if (kF16Terms == 3) bl[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(buf + (tt * FR + KS + s) * 1024 + lane * 16));
        }
        const int col = (t0 + tt) * 32 + r32;
        const float nj = *reinterpret_cast<const float*>(buf + ST * FR * 1024 + (tt * 32 + r32) * 4);     // |i_col|
#pragma unroll
        for (int ub = 0; ub < UB; ++ub) {
          floatx16 acc;
#pragma unroll
          for (int t = 0; t < 16; ++t) acc[t] = -thr[ub][t];                 // (the slab stage: thr = 0)
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            if (kF16Terms >= 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ub][s], bh[s], acc, 0, 0, 0);
            if (kF16Terms == 3) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ub][s], bl[s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ub][s], bh[s], acc, 0, 0, 0);
          }
          if (SLAB) {
            // a LOWER bound of every exact score of the slice: the bound stage ranks these
#pragma unroll
            for (int t = 0; t < 16; ++t) {
              const int row = m0 + 32 * ub + (t & 3) + 8 * (t >> 2) + 4 * h;
              if (row < m && col < n) C[(size_t)row * n + col] = __builtin_fmaf(-cu[ub][t], nj, acc[t]);
            }
            continue;
          }
#if SRH_F16_EXP == 2                          // (timing experiment: MFMAs + operand traffic + barriers alone)
          {
            float mx = acc[0];
#pragma unroll
            for (int t = 1; t < 16; ++t) mx = fmaxf(mx, acc[t]);
            if (mx == 12345.678f) f.cand_sc[0] = mx;
            continue;
          }
#endif
          // Nothing passes in one block of five: that case is straight-line code -- one FMA per output (the score's own
          // margin added to s~ - T), their maximum by v_max3, ONE comparison and ONE ballot per 32 x 32 block.  (Round 3 packed
          // 16 comparisons into a bit mask first: two VALU operations per output; its first version tested a ballot per
          // accumulator register: 32 taken branches per tile.)
          float e[16];
#pragma unroll
          for (int t = 0; t < 16; ++t) e[t] = __builtin_fmaf(cu[ub][t], nj, acc[t]);
          float mx = fmaxf(fmaxf(e[0], e[1]), e[2]);
#pragma unroll
          for (int t = 3; t < 15; t += 2) mx = fmaxf(fmaxf(mx, e[t]), e[t + 1]);
          mx = fmaxf(mx, e[15]);
          if (col >= n) mx = -1.f;
          unsigned long long bal = __builtin_amdgcn_ballot_w64(mx >= 0.f);
          if (bal == 0) continue;
          unsigned pm = 0;
#pragma unroll
          for (int t = 0; t < 16; ++t) pm |= (e[t] >= 0.f ? 1u : 0u) << t;           // (rows >= m: -inf)
          if (col >= n) pm = 0;
          bal = __builtin_amdgcn_ballot_w64(pm != 0);
          while (bal != 0) {                                                 // wave-uniform; one pass per survivor of the
            const bool act = pm != 0;                                        // lane that has the most (almost always 1)
            const int t = act ? __builtin_ctz(pm) : 0;
            pm &= pm - 1u;
            float sc = acc[0] + thr[ub][0];
#pragma unroll
            for (int k = 1; k < 16; ++k) sc = (t == k) ? acc[k] + thr[ub][k] : sc;       // (registers cannot be indexed by t)
            if (act) {
              const int at = staged + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0));
              stage_row[at] = (short)(32 * ub + (t & 3) + 8 * (t >> 2) + 4 * h);
              stage_col[at] = col;
              stage_sc[at] = sc;
            }
            staged += __builtin_popcountll(bal);
            if (staged > CAP - 64) flush();                                 // (rare: a tie-heavy block)
            bal = __builtin_amdgcn_ballot_w64(pm != 0);
          }
        }
      }
      if (staged > 0) flush();      // once per stage, right before the wait the pipeline makes anyway
    }
  }
}

// Survivors of filter16_kernel -> the row's exact top-K.  A survivor carries its bf16 score s~ with |s~ - s| <= delta_j =
// c |u| |i_j| (+ the rounding of handing it on); only the few that can still be in the exact top-K are re-scored:
//   L_j = s~ - delta_j <= s_j <= U_j = s~ + delta_j;  tau = the K-th largest L_j of the unmasked survivors: K items have
//   s >= tau, so the exact K-th best is >= tau and an exact top-K item has U_j >= tau -- the candidates of the second round
//   (K ... a few dozen per user: their item rows are the only ones gathered again).
// The exact score is the fma chain  acc = fma(u[s], i[s], acc); acc = fma(u[D/2 + s], i[D/2 + s], acc), s = 0 .. D/2 - 1 --
// BIT-IDENTICAL to what v_mfma_f32_32x32x2_f32 accumulates in gemm_nt_kernel with its operand assignment (measured on
// gfx950: tools/microbench/mfma_chain.hip, 40 / 40 tiles; every other order, 0 / 40), so ids and scores equal the plain
// pipeline's bit for bit (tests/test_gpu_kernels.py compares them).  One workgroup per query row; rows whose list
// overflowed are left alone.
template <int D>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void rescore_topk_kernel(const float* __restrict__ U, const int32_t* __restrict__ user_ids,
                                                           int user_base, const float* __restrict__ I,
                                                           const int32_t* __restrict__ cnt, const int32_t* __restrict__ cand_id,
                                                           const float* __restrict__ cand_sc, const float* __restrict__ u_norm,
                                                           const float* __restrict__ item_norm,
                                                           const float* __restrict__ thr, int thr_stride,
                                                           int cap, int k, int bitmap_words,
                                                           const int32_t* __restrict__ r_indptr,
                                                           const int32_t* __restrict__ r_indices,
                                                           int32_t* __restrict__ out_ids, float* __restrict__ out_scores,
                                                           int skip_upto) {
  constexpr int DH = D / 2;
  extern __shared__ unsigned char cand_smem[];
  float* s_val = reinterpret_cast<float*>(cand_smem);         // approximate scores, then exact ones of the second round
  int* s_idx = reinterpret_cast<int*>(cand_smem) + cap;
  int* s_sel = s_idx + cap;                                    // second-round candidates (positions in s_idx)
  float* s_dl = reinterpret_cast<float*>(s_sel + cap);         // the survivors' margins: U_j - L_j
  __shared__ float s_u[D];
  __shared__ int s_n, s_m;
  __shared__ float s_tau;
  const int row = blockIdx.x;
  const int c = cnt[row];
  if (c > cap || c <= skip_upto) return;                       // (short lists: rescore_wave_kernel has ranked them)
  if (threadIdx.x == 0) { s_n = 0; s_m = 0; s_tau = -INFINITY; }
  const int u = user_ids ? user_ids[row] : user_base + row;          // the user's id: mask CSR row
  const int urow = user_ids ? u : row;
  if (threadIdx.x < D) s_u[threadIdx.x] = U[(size_t)urow * D + threadIdx.x];
  __syncthreads();
  // training items of the user are dropped here (graph_recommender.py:49-50 masks them to -10e8)
  const int rs = r_indptr ? r_indptr[u] : 0, re = r_indptr ? r_indptr[u + 1] : 0;
  // (compaction by wave ballot: one LDS atomic per wave and round -- 370 same-address LDS atomics per row, one per
  // candidate, were a third of this kernel's time)
  const int lane = threadIdx.x & 63;
  // membership in the user's training row: a bitmap over the catalogue in LDS (bitmap_words > 0: catalogues up to 131 k
  // items) filled from the row with one coalesced pass -- a binary search per candidate is a chain of 6-7 dependent L2
  // round trips, ~10 us of this kernel's 40 per workgroup -- or the binary search for larger catalogues
  unsigned int* s_bits = reinterpret_cast<unsigned int*>(s_dl + cap);
  // delta_j = cu |i_j| + slack: cu = c |u|; the slack covers the rounding of s~ = (s~ - T) + T in the filter's hand-off
  const float cu = kFilterMargin * u_norm[row];
  const float slack = kFilterAbsSlack * fabsf(thr[(size_t)row * thr_stride]);
  if (bitmap_words > 0) {
    for (int w = threadIdx.x; w < bitmap_words; w += 256) s_bits[w] = 0u;
    __syncthreads();
    for (int p = rs + threadIdx.x; p < re; p += 256) {
      const int item = r_indices[p];
      atomicOr(&s_bits[item >> 5], 1u << (item & 31));
    }
    __syncthreads();
  }
  for (int t0 = 0; t0 < c; t0 += 256) {
    const int t = t0 + threadIdx.x;
    bool keep = t < c;
    int id = 0;
    if (keep) {
      id = cand_id[(size_t)row * cap + t];
      if (bitmap_words > 0) {
        keep = !((s_bits[id >> 5] >> (id & 31)) & 1u);
      } else {
        int lo = rs, hi = re;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (r_indices[mid] < id) lo = mid + 1; else hi = mid;
        }
        keep = !(lo < re && r_indices[lo] == id);
      }
    }
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(keep);
    int base = 0;
    if (lane == 0 && bal) base = atomicAdd(&s_n, __builtin_popcountll(bal));
    base = __builtin_amdgcn_readfirstlane(base);
    if (keep) {
      const int at = base + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0));
      const float sc = cand_sc[(size_t)row * cap + t];
      const float dl = __builtin_fmaf(cu, item_norm[id], __builtin_fmaf(kFilterAbsSlack, fabsf(sc), slack));
      s_val[at] = sc - dl;                                     // L_j (rounded down by the slack's share)
      s_dl[at] = 2.0f * dl;                                    // U_j = L_j + 2 delta_j
      s_idx[at] = id;
    }
  }
  __syncthreads();
  const int nv = s_n;
  // tau: the lower bound of rank min(K, nv) - 1 (fewer than K survivors: everything is re-scored)
  for (int t = threadIdx.x; t < nv; t += 256) {
    const float v = s_val[t];
    const int id = s_idx[t];
    int rank = 0;
    for (int j = 0; j < nv; ++j) rank += (s_val[j] > v) || (s_val[j] == v && s_idx[j] < id);
    if (rank == min(k, nv) - 1) s_tau = (nv >= k) ? v : -INFINITY;
  }
  __syncthreads();
  const float tau = s_tau;
  for (int t0 = 0; t0 < nv; t0 += 256) {
    const int t = t0 + threadIdx.x;
    const bool keep = t < nv && s_val[t] + s_dl[t] >= tau;
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(keep);
    int base = 0;
    if (lane == 0 && bal) base = atomicAdd(&s_m, __builtin_popcountll(bal));
    base = __builtin_amdgcn_readfirstlane(base);
    if (keep) s_sel[base + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0))] = t;
  }
  __syncthreads();
  const int ns = s_m;
  // exact scores of the second round, in place
  for (int e = threadIdx.x; e < ns; e += 256) {
    const int t = s_sel[e];
    const float4* ip = reinterpret_cast<const float4*>(I + (size_t)s_idx[t] * D);
    float acc = 0.f;
#pragma unroll 4
    for (int q = 0; q < DH / 4; ++q) {
      const float4 x0 = ip[q], x1 = ip[DH / 4 + q];
      acc = __builtin_fmaf(s_u[4 * q + 0], x0.x, acc); acc = __builtin_fmaf(s_u[DH + 4 * q + 0], x1.x, acc);
      acc = __builtin_fmaf(s_u[4 * q + 1], x0.y, acc); acc = __builtin_fmaf(s_u[DH + 4 * q + 1], x1.y, acc);
      acc = __builtin_fmaf(s_u[4 * q + 2], x0.z, acc); acc = __builtin_fmaf(s_u[DH + 4 * q + 2], x1.z, acc);
      acc = __builtin_fmaf(s_u[4 * q + 3], x0.w, acc); acc = __builtin_fmaf(s_u[DH + 4 * q + 3], x1.w, acc);
    }
    s_val[t] = acc;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < ns; e += 256) {
    const int t = s_sel[e];
    const float v = s_val[t];
    const int id = s_idx[t];
    int rank = 0;
    for (int j = 0; j < ns; ++j) {
      const int tj = s_sel[j];
      rank += (s_val[tj] > v) || (s_val[tj] == v && s_idx[tj] < id);
    }
    if (rank < k) {
      out_ids[(size_t)row * k + rank] = id;
      out_scores[(size_t)row * k + rank] = v;
    }
  }
}

// The same re-score with ONE WAVE per row, for the rows whose list is short (<= CW survivors: all but a few per cent).
// rescore_topk_kernel spends a 256-thread workgroup, seven barriers and a catalogue-wide bitmap on ~60 survivors: its
// 20 us per row are a chain of dependent round trips with seven rows in flight per CU.  Here a row is a wave: no barrier
// (a wave's LDS operations complete in order), the counters of the compactions are wave-uniform registers, membership in
// the user's training row is a binary search in an LDS copy of that row (rows longer than TRW: in global memory), and
// 24 rows are in flight per CU.  Same arithmetic, same order of the exact fma chain, same (score desc, id asc) ranking.
template <int D, int CW>
__global__ __launch_bounds__(256) void rescore_wave_kernel(const float* __restrict__ U, const int32_t* __restrict__ user_ids, int user_base,
                                                           const float* __restrict__ I, int rows, const int32_t* __restrict__ cnt,
                                                           const int32_t* __restrict__ cand_id, const float* __restrict__ cand_sc,
                                                           const float* __restrict__ u_norm, const float* __restrict__ item_norm,
                                                           const float* __restrict__ thr, int thr_stride, int cap, int k,
                                                           const int32_t* __restrict__ r_indptr, const int32_t* __restrict__ r_indices,
                                                           int32_t* __restrict__ out_ids, float* __restrict__ out_scores) {
  constexpr int DH = D / 2, TRW = 512;
  __shared__ float s_val_[4][CW];
  __shared__ float s_dl_[4][CW];
  __shared__ int s_idx_[4][CW];
  __shared__ short s_sel_[4][CW];
  __shared__ int s_tr_[4][TRW];
  __shared__ float s_u_[4][D];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + w;
  if (row >= rows) return;
  const int c = cnt[row];
  if (c > CW) return;                                             // (long or overflowed lists: rescore_topk_kernel)
  float* s_val = s_val_[w];
  float* s_dl = s_dl_[w];
  int* s_idx = s_idx_[w];
  short* s_sel = s_sel_[w];
  int* s_tr = s_tr_[w];
  float* s_u = s_u_[w];
  const int u = user_ids ? user_ids[row] : user_base + row;
  const int urow = user_ids ? u : row;
  for (int q = lane; q < D; q += 64) s_u[q] = U[(size_t)urow * D + q];
  const int rs = r_indptr ? r_indptr[u] : 0, re = r_indptr ? r_indptr[u + 1] : 0;
  const int ntr = re - rs;
  const bool cached = ntr <= TRW;
  if (cached) for (int p = lane; p < ntr; p += 64) s_tr[p] = r_indices[rs + p];
  const float cu = kFilterMargin * u_norm[row];
  const float slack = kFilterAbsSlack * fabsf(thr[(size_t)row * thr_stride]);
  __builtin_amdgcn_wave_barrier();
  // 1. the unmasked survivors with their bounds: L_j in s_val, U_j - L_j in s_dl
  int nv = 0;                                                     // wave-uniform
  for (int t0 = 0; t0 < c; t0 += 64) {
    const int t = t0 + lane;
    bool keep = t < c;
    int id = 0;
    if (keep) {
      id = cand_id[(size_t)row * cap + t];
      int lo = 0, hi = ntr;
      if (cached) {
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (s_tr[mid] < id) lo = mid + 1; else hi = mid;
        }
        keep = !(lo < ntr && s_tr[lo] == id);
      } else {
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (r_indices[rs + mid] < id) lo = mid + 1; else hi = mid;
        }
        keep = !(lo < ntr && r_indices[rs + lo] == id);
      }
    }
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(keep);
    if (keep) {
      const int at = nv + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0));
      const float sc = cand_sc[(size_t)row * cap + t];
      const float dl = __builtin_fmaf(cu, item_norm[id], __builtin_fmaf(kFilterAbsSlack, fabsf(sc), slack));
      s_val[at] = sc - dl;
      s_dl[at] = 2.0f * dl;
      s_idx[at] = id;
    }
    nv += __builtin_popcountll(bal);
  }
  __builtin_amdgcn_wave_barrier();
  // 2. tau: the lower bound of rank min(K, nv) - 1 (fewer than K survivors: everything is re-scored)
  float tau = -INFINITY;
  if (nv >= k) {
    for (int t0 = 0; t0 < nv; t0 += 64) {
      const int t = t0 + lane;
      bool hit = false;
      float v = 0.f;
      if (t < nv) {
        v = s_val[t];
        const int id = s_idx[t];
        int rank = 0;
        for (int j = 0; j < nv; ++j) rank += (s_val[j] > v) || (s_val[j] == v && s_idx[j] < id);
        hit = rank == k - 1;
      }
      const unsigned long long bal = __builtin_amdgcn_ballot_w64(hit);
      if (bal) tau = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (int)__builtin_ctzll(bal)));
    }
  }
  // 3. the candidates of the second round: U_j >= tau
  int ns = 0;                                                     // wave-uniform
  for (int t0 = 0; t0 < nv; t0 += 64) {
    const int t = t0 + lane;
    const bool keep = t < nv && s_val[t] + s_dl[t] >= tau;
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(keep);
    if (keep) s_sel[ns + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0))] = (short)t;
    ns += __builtin_popcountll(bal);
  }
  __builtin_amdgcn_wave_barrier();
  // 4. their exact scores, in place: a lane per candidate, the fma chain of gemm_nt_kernel's MFMA in its order
  for (int e = lane; e < ns; e += 64) {
    const int t = s_sel[e];
    const float4* ip = reinterpret_cast<const float4*>(I + (size_t)s_idx[t] * D);
    float acc = 0.f;
#pragma unroll 4
    for (int q = 0; q < DH / 4; ++q) {
      const float4 x0 = ip[q], x1 = ip[DH / 4 + q];
      acc = __builtin_fmaf(s_u[4 * q + 0], x0.x, acc); acc = __builtin_fmaf(s_u[DH + 4 * q + 0], x1.x, acc);
      acc = __builtin_fmaf(s_u[4 * q + 1], x0.y, acc); acc = __builtin_fmaf(s_u[DH + 4 * q + 1], x1.y, acc);
      acc = __builtin_fmaf(s_u[4 * q + 2], x0.z, acc); acc = __builtin_fmaf(s_u[DH + 4 * q + 2], x1.z, acc);
      acc = __builtin_fmaf(s_u[4 * q + 3], x0.w, acc); acc = __builtin_fmaf(s_u[DH + 4 * q + 3], x1.w, acc);
    }
    s_val[t] = acc;
  }
  __builtin_amdgcn_wave_barrier();
  // 5. (score desc, id asc) among them
  for (int e = lane; e < ns; e += 64) {
    const int t = s_sel[e];
    const float v = s_val[t];
    const int id = s_idx[t];
    int rank = 0;
    for (int j = 0; j < ns; ++j) {
      const int tj = s_sel[j];
      rank += (s_val[tj] > v) || (s_val[tj] == v && s_idx[tj] < id);
    }
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
// The bound stage needs ONE number per row: a lower bound of the K-th largest score of the (masked) sample slab.  One wave per
// row: every lane keeps the maxima of four interleaved groups of the row (256 disjoint groups in all), then K rounds of
// "wave maximum, remove it once": the K-th largest of 256 maxima of disjoint groups is attained by K distinct elements, so it
// is a lower bound of the row's K-th largest element (equal to it unless two of the row's top K share a group).  One pass over
// the slab at the fabric's rate instead of topk_kernel's two passes + candidate ranking (152 -> 64 us per 16384 x 4096 slab).
template <int CTRL>
__device__ __forceinline__ float dpp_max(float v) {
  const int iv = __float_as_int(v);
  return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(iv, iv, CTRL, 0xf, 0xf, false)));
}
__device__ __forceinline__ float wave_max_f(float v) {
  v = dpp_max<0x121>(v); v = dpp_max<0x122>(v); v = dpp_max<0x124>(v); v = dpp_max<0x128>(v);     // row_ror 1, 2, 4, 8
  const int iv = __float_as_int(v);
  return fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(iv, 0)), __int_as_float(__builtin_amdgcn_readlane(iv, 16))),
               fmaxf(__int_as_float(__builtin_amdgcn_readlane(iv, 32)), __int_as_float(__builtin_amdgcn_readlane(iv, 48))));
}

// mask_indptr != NULL: the row's wave first writes -10e8 over the user's training items in the slice (mask_kernel's job:
// graph_recommender.py:49-50) -- one launch and one pass over the masks less per chunk; the wave reads its own stores back
// after they have left it (s_waitcnt vmcnt(0): the vector L1 is write-through and holds none of these lines yet).
__global__ __launch_bounds__(256) void bound_rows_kernel(float* __restrict__ scores, int rows, int n, int k,
                                                         float* __restrict__ out, int out_stride,
                                                         const int32_t* __restrict__ user_ids, int user_base,
                                                         const int32_t* __restrict__ mask_indptr,
                                                         const int32_t* __restrict__ mask_indices,
                                                         const int32_t* __restrict__ column_of_item) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  if (mask_indptr) {
    const int u = user_ids ? user_ids[row] : user_base + row;
    const int ms = mask_indptr[u], me = mask_indptr[u + 1];
    for (int p = ms + lane; p < me; p += 64) {
      const int item = column_of_item[mask_indices[p]];           // (the slab's columns are image rows: items by norm)
      if (item < n) scores[(size_t)row * n + item] = -10e8f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const float* x = scores + (size_t)row * n;
  const int n4 = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) ? n / 4 : 0;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  float g0 = -INFINITY, g1 = -INFINITY, g2 = -INFINITY, g3 = -INFINITY;
  auto m4 = [](float4 v) { return fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)); };
  int i = lane;
  for (; i + 192 < n4; i += 256) {
    const float4 a = x4[i], b = x4[i + 64], c = x4[i + 128], d = x4[i + 192];
    g0 = fmaxf(g0, m4(a)); g1 = fmaxf(g1, m4(b)); g2 = fmaxf(g2, m4(c)); g3 = fmaxf(g3, m4(d));
  }
  for (; i < n4; i += 64) g0 = fmaxf(g0, m4(x4[i]));                  // (groups stay disjoint whatever their sizes)
  for (int j = 4 * n4 + lane; j < n; j += 64) g1 = fmaxf(g1, x[j]);
  float kth = -INFINITY;
  for (int r = 0; r < k; ++r) {                                       // k <= 128 < 256 groups
    const float mine = fmaxf(fmaxf(g0, g1), fmaxf(g2, g3));
    kth = wave_max_f(mine);
    const unsigned long long has = __builtin_amdgcn_ballot_w64(mine == kth);
    if (has == 0) break;                                              // (NaN scores: no lane equals the maximum)
    if (lane == (int)__builtin_ctzll(has)) {
      if (g0 == kth) g0 = -INFINITY; else if (g1 == kth) g1 = -INFINITY; else if (g2 == kth) g2 = -INFINITY; else g3 = -INFINITY;
    }
  }
  if (lane == 0) out[(size_t)row * out_stride] = kth;
}

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
// (rows, k + 1) ranked ids / scores -> their first k columns, with a mark on the rows in which two NEIGHBOURS of the k + 1
// scores are equal: ids_out[row][0] = -1 - id.  Such a row's order is the reference's heap walk's, not (score, id)'s
// (util/algorithm.py:144-156): the caller redoes it on the host.  One thread per row.
__global__ __launch_bounds__(256) void trim_mark_ties_kernel(const int32_t* __restrict__ ids, const float* __restrict__ sc,
                                                             int64_t rows, int k1, int32_t* __restrict__ ids_out,
                                                             float* __restrict__ sc_out) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  const int k = k1 - 1;
  bool tie = false;
  float prev = sc[r * k1];
  for (int c = 1; c < k1; ++c) {
    const float v = sc[r * k1 + c];
    tie |= (v == prev);
    prev = v;
  }
  for (int c = 0; c < k; ++c) {
    ids_out[r * k + c] = ids[r * k1 + c];
    sc_out[r * k + c] = sc[r * k1 + c];
  }
  if (tie) ids_out[r * k] = -1 - ids_out[r * k];
}

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
// (f-3) per-user hits and DCG / IDCG at up to 8 cut-offs: one thread per user, float64 adds in position order (the adds
// of Metric.NDCG's generator sum; a non-hit adds nothing there and +0.0 here)
struct MetricCuts { int32_t n; int32_t cut[8]; };
__global__ __launch_bounds__(256) void metric_rows_kernel(const uint8_t* __restrict__ flags, const int32_t* __restrict__ sizes,
                                                          int64_t n_query, int k, MetricCuts cuts,
                                                          const double* __restrict__ gains, const double* __restrict__ ideal,
                                                          int32_t* __restrict__ hits, double* __restrict__ ndcg) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= n_query) return;
  const uint8_t* f = flags + q * k;
  const int size = sizes[q];
  for (int c = 0; c < cuts.n; ++c) {
    const int n = min(cuts.cut[c], k);
    int h = 0;
    double dcg = 0.0;
    for (int pos = 0; pos < n; ++pos) {
      const int hit = f[pos] != 0;
      h += hit;
      dcg = dcg + (hit ? gains[pos] : 0.0);
    }
    hits[(size_t)c * n_query + q] = h;
    ndcg[(size_t)c * n_query + q] = dcg / ideal[(size_t)c * (k + 1) + min(size, cuts.cut[c])];
  }
}
}  // namespace

extern "C" {

srh_status_t srh_metric_rows(const uint8_t* d_flags, const int32_t* d_sizes, int64_t n_query, int32_t k,
                             const int32_t* cuts, int32_t n_cuts, const double* d_gains, const double* d_ideal,
                             int32_t* d_hits, double* d_ndcg, void* stream) {
  SRH_REQUIRE(d_flags && d_sizes && cuts && d_gains && d_ideal && d_hits && d_ndcg, "metric_rows: null argument");
  SRH_REQUIRE(n_query > 0 && k >= 1 && n_cuts >= 1 && n_cuts <= 8, "metric_rows: bad shape (1..8 cut-offs)");
  MetricCuts mc{};
  mc.n = n_cuts;
  for (int c = 0; c < n_cuts; ++c) {
    SRH_REQUIRE(cuts[c] >= 1 && cuts[c] <= k, "metric_rows: cut-off %d outside [1, k = %d]", cuts[c], k);
    mc.cut[c] = cuts[c];
  }
  metric_rows_kernel<<<(unsigned)((n_query + 255) / 256), 256, 0, srh::as_stream(stream)>>>(d_flags, d_sizes, n_query, k, mc,
                                                                                              d_gains, d_ideal, d_hits, d_ndcg);
  SRH_LAUNCH_CHECK();
  return SRH_OK;
}

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
  const int64_t padded = (n_items + 31) / 32 * 32;               // the item images are whole 32-row tiles
  // + the norm order of the catalogue: norms by item (the sort's keys), sorted keys, 0 .. n-1, the order, its inverse, the
  //   sort's scratch
  return 2 * filt_align(padded * d * 2) + 2 * filt_align(rows * d * 2) + filt_align(rows * 4) + filt_align(padded * 4) +
         5 * filt_align(padded * 4) + filt_align((int64_t)filt_sort_temp_bytes(n_items));
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
  float *u_norm = nullptr, *i_norm = nullptr;
  const float* norm_by_item = nullptr;
  int32_t *order = nullptr, *column_of_item = nullptr;
  if (split) {
    ws += filt_align(chunk_rows * 4);            // (the chunk layout's counter slot: counts live in d_out_counts)
    const int64_t padded = (n_items + 31) / 32 * 32;
    i_hi = reinterpret_cast<uint16_t*>(ws); ws += 2 * filt_align(padded * d * 2);      // hi and lo fragments, tile by tile
    i_lo = nullptr;
    u_hi = reinterpret_cast<uint16_t*>(ws); ws += filt_align(chunk_rows * d * 2);
    u_lo = reinterpret_cast<uint16_t*>(ws); ws += filt_align(chunk_rows * d * 2);
    u_norm = reinterpret_cast<float*>(ws); ws += filt_align(chunk_rows * 4);
    i_norm = reinterpret_cast<float*>(ws); ws += filt_align(padded * 4);
    uint32_t* norm_bits = reinterpret_cast<uint32_t*>(ws); ws += filt_align(padded * 4);       // = norms by ITEM, as floats
    uint32_t* sorted_bits = reinterpret_cast<uint32_t*>(ws); ws += filt_align(padded * 4);
    int32_t* iota = reinterpret_cast<int32_t*>(ws); ws += filt_align(padded * 4);
    order = reinterpret_cast<int32_t*>(ws); ws += filt_align(padded * 4);
    column_of_item = reinterpret_cast<int32_t*>(ws); ws += filt_align(padded * 4);
    void* sort_temp = ws;
    size_t sort_bytes = filt_sort_temp_bytes(n_items);
    norm_by_item = reinterpret_cast<const float*>(norm_bits);
    // (the last tile's rows beyond the catalogue: norm 0 -- their scores are never looked at)
    hipError_t err = padded > n_items ? hipMemsetAsync(i_norm + n_items, 0, sizeof(float) * (size_t)(padded - n_items), st) : hipSuccess;
    if (err != hipSuccess) { srh::set_error("score_mask_topk_filtered: %s", hipGetErrorString(err)); return SRH_ERR_HIP; }
    const int lpr = d / 4, g = 64 / lpr;
    const int blocks = (int)(((n_items + g - 1) / g + 3) / 4);
    // the catalogue in norm order, largest first (see item_norms_kernel), then its operand image in that order
    if (d == 64) item_norms_kernel<16><<<blocks, 256, 0, st>>>(d_item_emb, (int)n_items, norm_bits, iota);
    else item_norms_kernel<32><<<blocks, 256, 0, st>>>(d_item_emb, (int)n_items, norm_bits, iota);
    SRH_LAUNCH_CHECK();
    err = hipcub::DeviceRadixSort::SortPairsDescending(sort_temp, sort_bytes, norm_bits, sorted_bits, iota, order, (int)n_items, 0, 32, st);
    if (err != hipSuccess) { srh::set_error("score_mask_topk_filtered: sort: %s", hipGetErrorString(err)); return SRH_ERR_HIP; }
    invert_order_kernel<<<(int)((n_items + 255) / 256), 256, 0, st>>>(order, (int)n_items, column_of_item);
    SRH_LAUNCH_CHECK();
    if (d == 64) split_rows_kernel<16, true><<<blocks, 256, 0, st>>>(d_item_emb, order, (int)n_items, i_hi, i_lo, i_norm, nullptr);
    else split_rows_kernel<32, true><<<blocks, 256, 0, st>>>(d_item_emb, order, (int)n_items, i_hi, i_lo, i_norm, nullptr);
    SRH_LAUNCH_CHECK();
    static const bool attr_set = [] {
      (void)hipFuncSetAttribute((const void*)filter16_kernel<64, false, SRH_F16_UB>, hipFuncAttributeMaxDynamicSharedMemorySize, kF16Lds);
      (void)hipFuncSetAttribute((const void*)filter16_kernel<128, false, SRH_F16_UB>, hipFuncAttributeMaxDynamicSharedMemorySize, kF16Lds);
      (void)hipFuncSetAttribute((const void*)filter16_kernel<64, true, SRH_F16_UB>, hipFuncAttributeMaxDynamicSharedMemorySize, kF16Lds);
      (void)hipFuncSetAttribute((const void*)filter16_kernel<128, true, SRH_F16_UB>, hipFuncAttributeMaxDynamicSharedMemorySize, kF16Lds);
      return true;
    }();
    (void)attr_set;
  }
  for (int64_t lo = 0; lo < n_query; lo += chunk_rows) {
    const int64_t m = std::min(chunk_rows, n_query - lo);
    const float* emb = d_user_ids ? d_user_emb : d_user_emb + lo * d;
    const int32_t* ids = d_user_ids ? d_user_ids + lo : nullptr;
    int32_t* cnt = d_out_counts + lo;
    // 1. top-K over a leading slice of the catalogue: its K-th score is a lower bound of the
    //    row's overall K-th score (a subset's K-th best cannot beat the whole set's).  Exact f32 scores, or (split path)
    //    split-bf16 scores s~: the K-th largest exact score is >= the K-th largest s~ - delta_u, which the filter accounts for
    srh_status_t rc = SRH_OK;
    if (split) {
      const int lpr = d / 4, g = 64 / lpr;
      const int sb = (int)(((m + g - 1) / g + 3) / 4);
      const int s_tiles = (int)((sample_items + 31) / 32);
      const int gy = (int)((m + 255) / 256);
      const int gx = std::max(1, std::min(s_tiles, (512 + gy - 1) / gy));
      const int tpw = (s_tiles + gx - 1) / gx;
      dim3 grid((unsigned)((s_tiles + tpw - 1) / tpw), (unsigned)gy);
      Filter16Args none{};
      none.u_norm = u_norm;
      none.item_norm = i_norm;
      none.order = order;
      if (d == 64) {
        split_rows_kernel<16, false><<<sb, 256, 0, st>>>(emb, ids, (int)m, u_hi, u_lo, u_norm, nullptr);
        filter16_kernel<64, true, SRH_F16_UB><<<grid, 512 / SRH_F16_UB, kF16Lds, st>>>(u_hi, u_lo, i_hi, (int)m, (int)sample_items, none, slab);
      } else {
        split_rows_kernel<32, false><<<sb, 256, 0, st>>>(emb, ids, (int)m, u_hi, u_lo, u_norm, nullptr);
        filter16_kernel<128, true, SRH_F16_UB><<<grid, 512 / SRH_F16_UB, kF16Lds, st>>>(u_hi, u_lo, i_hi, (int)m, (int)sample_items, none, slab);
      }
      SRH_LAUNCH_CHECK();
    } else {
      rc = gemm_dispatch(emb, ids, d_item_emb, slab, m, sample_items, d, st);
      if (rc) return rc;
    }
    if (d_r_indptr && !split) {
      mask_kernel<<<(int)((m + 3) / 4), 256, 0, st>>>(ids, (int)m, d_r_indptr, d_r_indices, slab, (int)sample_items, (int)lo);
      SRH_LAUNCH_CHECK();
    }
    if (split) {
      // (the split path needs the bound only, not the sample's ranked list: one pass, one wave per row, the row's masks
      // applied by the same wave)
      bound_rows_kernel<<<(int)((m + 3) / 4), 256, 0, st>>>(slab, (int)m, (int)sample_items, k, s_sc + (k - 1), k, ids, (int)lo,
                                                            d_r_indptr, d_r_indices, column_of_item);
      SRH_LAUNCH_CHECK();
    } else {
      rc = srh_topk_rows(slab, m, sample_items, k, s_ids, s_sc, stream);
      if (rc) return rc;
    }
    // 2. all scores again, never stored: only those reaching the bound and not masked are kept
    hipError_t err = hipMemsetAsync(cnt, 0, sizeof(int32_t) * m, st);
    if (err != hipSuccess) { srh::set_error("score_mask_topk_filtered: %s", hipGetErrorString(err)); return SRH_ERR_HIP; }
    if (split) {
      // 2'. the filter on split-bf16 operands against the bound lowered by its error margins ...
      // 256 query rows per workgroup; the item tiles are dealt over enough workgroups to give every CU two
      const int n_tiles = (int)((n_items + 31) / 32);
      const int gy = (int)((m + 255) / 256);
      const int gx = std::max(1, std::min(n_tiles, (512 + gy - 1) / gy));
      const int tiles_per_wg = (n_tiles + gx - 1) / gx;
      dim3 grid((unsigned)((n_tiles + tiles_per_wg - 1) / tiles_per_wg), (unsigned)gy);
      Filter16Args f16{s_sc + (k - 1), k, u_norm, i_norm, order, cnt, cand_id, cand_sc, cap};
      if (d == 64) filter16_kernel<64, false, SRH_F16_UB><<<grid, 512 / SRH_F16_UB, kF16Lds, st>>>(u_hi, u_lo, i_hi, (int)m, (int)n_items, f16);
      else filter16_kernel<128, false, SRH_F16_UB><<<grid, 512 / SRH_F16_UB, kF16Lds, st>>>(u_hi, u_lo, i_hi, (int)m, (int)n_items, f16);
      SRH_LAUNCH_CHECK();
      // 3'. ... and the survivors re-scored by gemm_nt_kernel's own instruction sequence, masked, ranked
      // (training-row membership by an LDS bitmap over the catalogue while it fits: <= 16 KB, i.e. 131 k items)
      const int bitmap_words = (d_r_indptr && n_items <= 131072) ? (int)((n_items + 31) / 32) : 0;
      const size_t rs_lds = (size_t)cap * 16 + (size_t)bitmap_words * 4;
      // short lists (all but a few per cent of the rows) by a wave each, the rest by a workgroup each
      constexpr int kWaveRows = 128;               // (longer lists: the O(n^2) rank counts want the 256 threads of the workgroup form)
      const int wg = (int)((m + 3) / 4);
      if (d == 64) {
        rescore_wave_kernel<64, kWaveRows><<<wg, 256, 0, st>>>(emb, ids, (int)lo, d_item_emb, (int)m, cnt, cand_id, cand_sc, u_norm, norm_by_item,
                                                               s_sc + (k - 1), k, cap, k, d_r_indptr, d_r_indices,
                                                               d_out_ids + lo * k, d_out_scores + lo * k);
        rescore_topk_kernel<64><<<(int)m, 256, rs_lds, st>>>(emb, ids, (int)lo, d_item_emb, cnt, cand_id, cand_sc, u_norm,
                                                             norm_by_item, s_sc + (k - 1), k, cap, k, bitmap_words, d_r_indptr, d_r_indices,
                                                             d_out_ids + lo * k, d_out_scores + lo * k, kWaveRows);
      } else {
        rescore_wave_kernel<128, kWaveRows><<<wg, 256, 0, st>>>(emb, ids, (int)lo, d_item_emb, (int)m, cnt, cand_id, cand_sc, u_norm, norm_by_item,
                                                                s_sc + (k - 1), k, cap, k, d_r_indptr, d_r_indices,
                                                                d_out_ids + lo * k, d_out_scores + lo * k);
        rescore_topk_kernel<128><<<(int)m, 256, rs_lds, st>>>(emb, ids, (int)lo, d_item_emb, cnt, cand_id, cand_sc, u_norm,
                                                              norm_by_item, s_sc + (k - 1), k, cap, k, bitmap_words, d_r_indptr, d_r_indices,
                                                              d_out_ids + lo * k, d_out_scores + lo * k, kWaveRows);
      }
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

srh_status_t srh_topk_trim_mark_ties(const int32_t* d_ids, const float* d_scores, int64_t n_query, int32_t k1,
                                     int32_t* d_out_ids, float* d_out_scores, void* stream) {
  SRH_REQUIRE(d_ids && d_scores && d_out_ids && d_out_scores, "topk_trim_mark_ties: null argument");
  SRH_REQUIRE(n_query > 0 && k1 >= 2, "topk_trim_mark_ties: bad shape");
  trim_mark_ties_kernel<<<(unsigned)((n_query + 255) / 256), 256, 0, srh::as_stream(stream)>>>(d_ids, d_scores, n_query, k1,
                                                                                               d_out_ids, d_out_scores);
  SRH_LAUNCH_CHECK();
  return SRH_OK;
}

}  // extern "C"
