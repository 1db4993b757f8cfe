// (a-5..a-8) Batch losses of the LightGCN family, forward + backward fused:
//   * row gather + BPR + L2 regulariser   (reference XSimGCL.py:30-33, util/loss_torch.py:6-10,18-22)
//   * InfoNCE                               (reference util/loss_torch.py:35-50)
// These kernels touch only O(batch) rows; what matters is launch count and, for InfoNCE,
// the 4 x (2 n^2 d) flops of the similarity products, which run on the fp32 MFMA pipe
// (v_mfma_f32_16x16x4_f32: exact f32 fma chains, so the 1e-4 parity budget is untouched).
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace {
using namespace srh;

typedef float floatx4 __attribute__((ext_vector_type(4)));

// row += v for a (4 LPR)-float row held one float4 per lane by an LPR-lane row-group, with the float atomics laid out
// DENSELY.  The obvious form -- x, y, z, w of the lane's float4, one atomic each -- issues four instructions whose lanes
// hit every fourth dword of the row: each instruction becomes one sparse request per 64 bytes of the row, and the loss
// section's ~0.9 M gradient atomics then wait on the L2's atomic units (nce_finish_bpr2 19.0 us; 9.2 us with racy plain
// stores in their place: profiles/r02_f_loss_atomics.txt).  Here the row-group first transposes through 1 KB of LDS
// per wave so that instruction k covers dwords [k LPR, (k+1) LPR) of the row -- the same atomics in a quarter of the
// requests: 9.8 us, as fast as the plain stores.
template <int LPR>
__device__ __forceinline__ void atomic_add_row(float* row, float4 v, int sub, float4* scr /* this wave's 64 float4 */) {
  const int lane = threadIdx.x & 63;
  scr[lane] = v;                                                 // lane-linear: a row-group's row is contiguous
  const float* grp = reinterpret_cast<const float*>(scr + (lane - sub));
#pragma unroll
  for (int k = 0; k < 4; ++k) unsafeAtomicAdd(row + k * LPR + sub, grp[k * LPR + sub]);
}

// ---------------------------------------------------------------------------------------
// BPR: per-row pieces shared by the gathered (engine) and plain (drop-in) entry points
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void bpr_row(float pos, float neg, float& loss, float& coef) {
  const float x = pos - neg;
  const float sig = 1.0f / (1.0f + expf(-x));   // torch.sigmoid
  const float arg = 10e-6f + sig;               // loss_torch.py:9
  loss = -logf(arg);
  coef = -(sig * (1.0f - sig)) / arg;           // d loss / d x
}

struct BprArgs {
  const float *user, *item, *reg_user, *reg_item;
  const int32_t *u_idx, *i_idx, *j_idx;
  const int32_t* d_n_rows;
  int B;
  float reg_coef, loss_scale;
  int reg_include_neg;
  float *g_user, *g_item, *greg_user, *greg_item;
  double* losses;
  double* part;    // ws: 4 doubles per phase-1 workgroup {loss, |Ru|^2, |Rp|^2, |Rn|^2}
  float* coef;     // ws: B floats
  int n_blocks;
};

template <int LPR>
__device__ __forceinline__ void bpr_phase1_body(const BprArgs& a, const unsigned block_id) {
  constexpr int G = 64 / LPR;
  const int rows = a.d_n_rows ? min(*a.d_n_rows, a.B) : a.B;
  const int lane = threadIdx.x & 63, g = lane / LPR, sub = lane % LPR;
  const int b = (int)((block_id * 256u + threadIdx.x) >> 6) * G + g;
  const bool valid = b < rows;
  const int bu = valid ? a.u_idx[b] : 0, bi = valid ? a.i_idx[b] : 0, bj = valid ? a.j_idx[b] : 0;
  const float4 u = reinterpret_cast<const float4*>(a.user)[(size_t)bu * LPR + sub];
  const float4 p = reinterpret_cast<const float4*>(a.item)[(size_t)bi * LPR + sub];
  const float4 n = reinterpret_cast<const float4*>(a.item)[(size_t)bj * LPR + sub];
  const float pos = group_sum<LPR>(f4_dot(u, p));
  const float neg = group_sum<LPR>(f4_dot(u, n));
  float loss, coef;
  bpr_row(pos, neg, loss, coef);
  if (valid && sub == 0) a.coef[b] = coef;
  float4 ru = u, rp = p, rn = n;
  if (a.reg_user != a.user) ru = reinterpret_cast<const float4*>(a.reg_user)[(size_t)bu * LPR + sub];
  if (a.reg_item != a.item) {
    rp = reinterpret_cast<const float4*>(a.reg_item)[(size_t)bi * LPR + sub];
    rn = reinterpret_cast<const float4*>(a.reg_item)[(size_t)bj * LPR + sub];
  }
  // per-lane partials (each lane owns 4 of the d columns); loss counted once per row
  double l_part = (valid && sub == 0) ? (double)loss : 0.0;
  double su = valid ? (double)f4_dot(ru, ru) : 0.0;
  double sp = valid ? (double)f4_dot(rp, rp) : 0.0;
  double sn = valid ? (double)f4_dot(rn, rn) : 0.0;
  // same-address atomics serialise at ~11 ns each on this chip (2048 of them cost 22 us):
  // reduce inside the workgroup and leave one partial record per workgroup instead
  __shared__ double s_part[4][4];
  l_part = wave_sum_d(l_part);
  su = wave_sum_d(su);
  sp = wave_sum_d(sp);
  sn = wave_sum_d(sn);
  const int wv = threadIdx.x >> 6;
  if (lane == 0) { s_part[wv][0] = l_part; s_part[wv][1] = su; s_part[wv][2] = sp; s_part[wv][3] = sn; }
  __syncthreads();
  if (threadIdx.x < 4)
    a.part[(size_t)block_id * 4 + threadIdx.x] =
        s_part[0][threadIdx.x] + s_part[1][threadIdx.x] + s_part[2][threadIdx.x] + s_part[3][threadIdx.x];
}

template <int LPR>
__device__ __forceinline__ void bpr_phase2_body(const BprArgs& a, const unsigned block_id) {
  constexpr int G = 64 / LPR;
  const int rows = a.d_n_rows ? min(*a.d_n_rows, a.B) : a.B;
  if (rows <= 0) return;
  const int lane = threadIdx.x & 63, g = lane / LPR, sub = lane % LPR;
  const int b = (int)((block_id * 256u + threadIdx.x) >> 6) * G + g;
  // the row data does not depend on the fold of phase 1's partials below: its loads go first (row-groups past the
  // batch re-read its last row and are dropped at the end)
  const int bq = min(b, rows - 1);
  const int bu = a.u_idx[bq], bi = a.i_idx[bq], bj = a.j_idx[bq];
  const float cf = a.coef[bq];
  const float4 u = reinterpret_cast<const float4*>(a.user)[(size_t)bu * LPR + sub];
  const float4 p = reinterpret_cast<const float4*>(a.item)[(size_t)bi * LPR + sub];
  const float4 n = reinterpret_cast<const float4*>(a.item)[(size_t)bj * LPR + sub];
  float4 ru = u, rp = p, rn = n;
  if (a.reg_user != a.user) ru = reinterpret_cast<const float4*>(a.reg_user)[(size_t)bu * LPR + sub];
  if (a.reg_item != a.item) {
    rp = reinterpret_cast<const float4*>(a.reg_item)[(size_t)bi * LPR + sub];
    rn = reinterpret_cast<const float4*>(a.reg_item)[(size_t)bj * LPR + sub];
  }
  __shared__ double s_tot[4];
  __shared__ float4 s_scr[4][64];
  float4* scr = s_scr[threadIdx.x >> 6];
  if (threadIdx.x < 64) {                 // wave 0 folds the per-workgroup partials (fixed order)
    double t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    for (int k = threadIdx.x; k < a.n_blocks; k += 64) {
      t0 += a.part[(size_t)k * 4 + 0]; t1 += a.part[(size_t)k * 4 + 1];
      t2 += a.part[(size_t)k * 4 + 2]; t3 += a.part[(size_t)k * 4 + 3];
    }
    t0 = wave_sum_d(t0); t1 = wave_sum_d(t1); t2 = wave_sum_d(t2); t3 = wave_sum_d(t3);
    if (threadIdx.x == 0) { s_tot[0] = t0; s_tot[1] = t1; s_tot[2] = t2; s_tot[3] = t3; }
  }
  __syncthreads();
  const float nu = (float)sqrt(s_tot[1]), np = (float)sqrt(s_tot[2]), nn = (float)sqrt(s_tot[3]);
  if (block_id == 0 && threadIdx.x == 0) {
    float r = nu / (float)rows + np / (float)rows;
    if (a.reg_include_neg) r += nn / (float)rows;
    a.losses[0] += (double)a.loss_scale * s_tot[0] / (double)rows;
    a.losses[1] += (double)(a.loss_scale * (r * a.reg_coef));
  }
  if (b >= rows) return;
  const float c = cf * (a.loss_scale / (float)rows);
  float4 gu = make_float4(c * (p.x - n.x), c * (p.y - n.y), c * (p.z - n.z), c * (p.w - n.w));
  float4 gp = f4_scale(u, c);
  float4 gn = f4_scale(u, -c);
  // d/dx ( reg_coef * ||X||_F / rows ) = reg_coef / rows * x / ||X||_F   (0 when ||X|| = 0)
  const float rs = a.reg_coef * a.loss_scale / (float)rows;
  const float cu = nu > 0.f ? rs / nu : 0.f, cp = np > 0.f ? rs / np : 0.f;
  const float cn = (a.reg_include_neg && nn > 0.f) ? rs / nn : 0.f;
  const bool same_u = (a.reg_user == a.user) && (a.greg_user == a.g_user);
  const bool same_i = (a.reg_item == a.item) && (a.greg_item == a.g_item);
  if (same_u) gu = f4_fma(cu, ru, gu);
  else atomic_add_row<LPR>(a.greg_user + (size_t)bu * LPR * 4, f4_scale(ru, cu), sub, scr);
  if (same_i) {
    gp = f4_fma(cp, rp, gp);
    gn = f4_fma(cn, rn, gn);
  } else {
    atomic_add_row<LPR>(a.greg_item + (size_t)bi * LPR * 4, f4_scale(rp, cp), sub, scr);
    if (a.reg_include_neg) atomic_add_row<LPR>(a.greg_item + (size_t)bj * LPR * 4, f4_scale(rn, cn), sub, scr);
  }
  atomic_add_row<LPR>(a.g_user + (size_t)bu * LPR * 4, gu, sub, scr);
  atomic_add_row<LPR>(a.g_item + (size_t)bi * LPR * 4, gp, sub, scr);
  atomic_add_row<LPR>(a.g_item + (size_t)bj * LPR * 4, gn, sub, scr);
}

template <int LPR>
__global__ __launch_bounds__(256) void bpr_phase1(BprArgs a) { bpr_phase1_body<LPR>(a, blockIdx.x); }
template <int LPR>
__global__ __launch_bounds__(256) void bpr_phase2(BprArgs a) { bpr_phase2_body<LPR>(a, blockIdx.x); }

// ---- the op-level tier's losses: every scalar is FINISHED ON THE DEVICE ------------------------------------------------
// A model file written the reference's way calls bpr_loss / l2_reg_loss once per step and gets a 0-dim tensor back; with
// torch ops around a partial-sum kernel that is a fill, a division, a conversion (forward) and a comparison, a where, two
// multiplies (backward) per call -- eight host dispatches for one number.  Here the LAST workgroup of the launch (a
// ticket in the caller's 64-byte scalar workspace) turns the double accumulators into the f32 result and leaves the
// workspace zero for the next call: one launch forward, one backward.
struct ScalarWs {
  double acc[4];
  unsigned int ticket;
  unsigned int pad[7];
};
static_assert(sizeof(ScalarWs) == SRH_SCALAR_WS_BYTES, "scalar workspace layout");

// block-wide sum handed to one atomic; returns true in thread 0 of the LAST block to arrive (all blocks' sums are then
// visible to it through device-scope atomics)
__device__ __forceinline__ bool block_acc_then_ticket(double part, double* acc, unsigned int* ticket, unsigned int n_blocks) {
  __shared__ double s_part[4];
  part = wave_sum_d(part);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x != 0) return false;
  atomicAdd(acc, (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]));
  __threadfence();
  return atomicAdd(ticket, 1u) == n_blocks - 1;
}

template <int LPR>
__global__ __launch_bounds__(256) void bpr_plain_fwd(const float4* __restrict__ U, const float4* __restrict__ P,
                                                     const float4* __restrict__ Nn, int B, ScalarWs* ws,
                                                     float* __restrict__ loss_mean, float* __restrict__ coef) {
  constexpr int G = 64 / LPR;
  const int lane = threadIdx.x & 63, g = lane / LPR, sub = lane % LPR;
  const int b = (int)((blockIdx.x * 256u + threadIdx.x) >> 6) * G + g;
  const bool valid = b < B;
  const size_t at = (size_t)(valid ? b : 0) * LPR + sub;
  const float4 u = U[at], p = P[at], n = Nn[at];
  const float pos = group_sum<LPR>(f4_dot(u, p));
  const float neg = group_sum<LPR>(f4_dot(u, n));
  float loss, c;
  bpr_row(pos, neg, loss, c);
  if (valid && sub == 0) coef[b] = c / (float)B;          // d mean / d (pos - neg) of row b
  if (block_acc_then_ticket((valid && sub == 0) ? (double)loss : 0.0, &ws->acc[0], &ws->ticket, gridDim.x)) {
    *loss_mean = (float)(atomicAdd(&ws->acc[0], 0.0) / (double)B);      // torch.mean(loss): loss_torch.py:10
    ws->acc[0] = 0.0;
    ws->ticket = 0u;
  }
}

template <int LPR>
__global__ __launch_bounds__(256) void bpr_plain_bwd(const float4* __restrict__ U, const float4* __restrict__ P,
                                                     const float4* __restrict__ Nn, const float* __restrict__ coef,
                                                     int B, const float* __restrict__ gout, float4* __restrict__ GU,
                                                     float4* __restrict__ GP, float4* __restrict__ GN) {
  const size_t t = (size_t)blockIdx.x * 256u + threadIdx.x;
  if (t >= (size_t)B * LPR) return;
  const int b = (int)(t / LPR);
  const float c = coef[b] * gout[0];                    // the upstream gradient is read on the device: no host sync
  const float4 u = U[t], p = P[t], n = Nn[t];
  GU[t] = make_float4(c * (p.x - n.x), c * (p.y - n.y), c * (p.z - n.z), c * (p.w - n.w));
  GP[t] = f4_scale(u, c);
  GN[t] = f4_scale(u, -c);
}

// l2_reg_loss(reg, *embs) = reg * sum_k ||emb_k||_F / rows_k (loss_torch.py:18-22), all blocks in one launch
constexpr int kL2Blocks = 4;
constexpr int kL2MaxWg = 1024;       // workgroups per block of rows
struct L2Args {
  const float* x[kL2Blocks];
  float* gx[kL2Blocks];
  int64_t n[kL2Blocks];              // elements
  float rows[kL2Blocks];
  int first_wg[kL2Blocks + 1];       // block k owns workgroups [first_wg[k], first_wg[k + 1])
  int count;
  float reg;
};

__global__ __launch_bounds__(256) void l2_reg_fwd_kernel(L2Args a, ScalarWs* ws, float* __restrict__ norms, float* __restrict__ loss) {
  int k = 0;
  while (k + 1 < a.count && (int)blockIdx.x >= a.first_wg[k + 1]) ++k;
  const int wg = (int)blockIdx.x - a.first_wg[k], n_wg = a.first_wg[k + 1] - a.first_wg[k];
  const float* __restrict__ x = a.x[k];
  const int64_t n = a.n[k], stride = (int64_t)n_wg * 256;
  double acc = 0.0;
  for (int64_t i = (int64_t)wg * 256 + threadIdx.x; i < n; i += stride) {
    const float v = x[i];
    acc += (double)v * (double)v;
  }
  __shared__ double s_part[4];
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x != 0) return;
  atomicAdd(&ws->acc[k], (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]));
  __threadfence();
  if (atomicAdd(&ws->ticket, 1u) != gridDim.x - 1) return;
  float emb_loss = 0.f;                                     // the reference's left-to-right float32 sum
  for (int j = 0; j < a.count; ++j) {
    const float nrm = (float)sqrt(atomicAdd(&ws->acc[j], 0.0));        // torch.norm(emb, p=2)
    norms[j] = nrm;
    emb_loss += nrm / a.rows[j];
    ws->acc[j] = 0.0;
  }
  *loss = emb_loss * a.reg;
  ws->ticket = 0u;
}

// d loss / d emb_k = emb_k * (((gout * reg) / rows_k) / ||emb_k||), 0 where the norm is 0 (torch's norm backward)
__global__ __launch_bounds__(256) void l2_reg_bwd_kernel(L2Args a, const float* __restrict__ norms, const float* __restrict__ gout) {
  int k = 0;
  while (k + 1 < a.count && (int)blockIdx.x >= a.first_wg[k + 1]) ++k;
  const int wg = (int)blockIdx.x - a.first_wg[k], n_wg = a.first_wg[k + 1] - a.first_wg[k];
  const float nrm = norms[k];
  const float coef = (nrm > 0.f) ? ((gout[0] * a.reg) / a.rows[k]) / nrm : 0.f;
  const float* __restrict__ x = a.x[k];
  float* __restrict__ g = a.gx[k];
  const int64_t n = a.n[k], stride = (int64_t)n_wg * 256;
  for (int64_t i = (int64_t)wg * 256 + threadIdx.x; i < n; i += stride) g[i] = x[i] * coef;
}

// ---------------------------------------------------------------------------------------
// InfoNCE
// ---------------------------------------------------------------------------------------
constexpr int kNceSplits = 16;      // workspace is sized for this many key splits
constexpr float kNceKqScale = 32.0f;                                  // K / Q images hold 32 x (see nce_prep_body)
constexpr float kNceInvKqScale2 = 1.0f / (kNceKqScale * kNceKqScale);   // the similarity product comes out 1024 x
// Two build-time shape constants of the tile passes (tools/spmm_lab/build_alt.sh builds the other settings for an A/B;
// no run-time knob ships):
//   SRH_NCE_SPLITS    key-range splits per query tile: 8 (256-key stages, one workgroup per CU) or 16 (128-key stages:
//                     two workgroups per CU with 3-term P.V products)
//   SRH_NCE_PV_TERMS  cross terms of the P.V product: 3 (bf16 hi/mid operands: 2^-18 per product) or 6 (hi/mid/lo:
//                     2^-27, f32-class; a third V image in the stage)
#ifndef SRH_NCE_SPLITS
#define SRH_NCE_SPLITS 8
#endif
#ifndef SRH_NCE_PV_TERMS
#define SRH_NCE_PV_TERMS 3
#endif
static_assert(SRH_NCE_SPLITS == 8 || SRH_NCE_SPLITS == 16, "SRH_NCE_SPLITS: 8 or 16");
static_assert(SRH_NCE_PV_TERMS == 3 || SRH_NCE_PV_TERMS == 6, "SRH_NCE_PV_TERMS: 3 or 6");
//   SRH_NCE_WT        1 (default): the tile passes' partial outputs (16.8 MB per step at n = 2048) leave with write-through
//                     stores: read next by the finish kernel on other XCDs, never again by their writers (-1.2 us per step)
#ifndef SRH_NCE_WT
#define SRH_NCE_WT 1
#endif
constexpr bool kNceWT = SRH_NCE_WT != 0;
constexpr int kNceUsedSplits = SRH_NCE_SPLITS;   // key-range splits per query tile (NceBatch::splits; the finish kernels unroll over it)
constexpr int kNcePvTerms = SRH_NCE_PV_TERMS;

// Arithmetic of the two n x n x d products (srh_infonce_set_precision).  The process DEFAULT is the reference's: exact f32
// multiply-adds on the f32 MFMA (SRH_NCE_F32; d = 256, which those passes do not serve, resolves to the split mode).
// SRH_NCE_SPLIT16 -- operands as short sums of 16-bit pieces on the 16-bit MFMA pipe: the similarity product on scaled f16
// hi + lo (logits to 2^-22), the P.V product on bf16 hi + mid [+ lo] -- is the faster opt-in (the mode is an argument of
// the multi-problem entry points; the environment variable SRH_NCE_SPLIT16 makes it the initial default).
std::atomic<int> g_nce_precision{getenv("SRH_NCE_SPLIT16") ? SRH_NCE_SPLIT16 : SRH_NCE_F32};

constexpr unsigned long long kLossUnreported = ~0ull;      // (see rows_finish: a loss-partial slot nobody has written yet)

struct NceWs {
  float *v1n, *v2n, *norm1, *norm2, *opart, *opart2, *lpart, *invl;
  float* ediag;         // split path: exp(s_ii / tau - 1 / tau) as pass 1's MFMA saw it (the pair's own weight is kept out of
                        // the accumulated sums: see nce_tile_lds)
  double* losspart;     // one partial per finish wave
  int32_t* ticket;      // np/16 + 1 arrival counters (zeroed by nce_prep, re-armed by the last arriver)
  struct NcePlan* plan; // problem 0's: the persistent passes' task list for this step (written by nce_prep)
  float* etile;         // all-f32 passes, n <= kNceReuseMax: pass 1's weights e_ij, stored [key j][query i] (row stride np),
                        // so that pass 2 reads its A fragments back instead of recomputing the logits (NULL: recompute)
  // split-bf16 operand images of the two normalised views (hi = bf16(x), lo = bf16(x - hi)):
  // both stored FRAGMENT-LINEAR: the 64 lanes of one MFMA operand load read one contiguous 1 KB
  //   kq_*[view]  [row/16][k-slice s][lane][8]: lane (c16 = lane&15, g = lane>>4) holds row 16*tile + c16,
  //               dims (D/4) g + 8 s + e -- K / Q operands of the similarity product
  //   vt_*[view]  [row/32][n-tile t][lane][8]: lane holds column NT*c16 + t of the 8 keys
  //               32*blk + {4g..4g+3, 16+4g..16+4g+3} -- V operand of the P.V product
  uint16_t *kq_hi[2], *kq_lo[2], *vt_hi[2], *vt_mid[2], *vt_lo[2];
  int64_t np;
  // the problem this workspace belongs to (several InfoNCE problems share each launch: blockIdx.z)
  const float *src1, *src2;
  const int32_t* idx;
  const int32_t* d_n;
  int n_max;
  float *g1, *g2;
  int g2_plain;         // g2's rows are this problem's alone: plain read-add-store
};
constexpr int kNceMaxProblems = 4;
struct NceBatch {
  NceWs w[kNceMaxProblems];
  int count;
  int splits;      // key-range splits per query tile (1..kNceSplits): the split-16 passes' fixed setting
  int slots;       // > 0: the tile passes are PERSISTENT -- `slots` workgroups share the (query block, key range) tasks of
                   // nce_plan, cut on the device from the live row counts (the all-f32 passes)
  int f32_only;    // the prep kernel leaves the 16-bit operand images unwritten
};

__host__ __device__ inline int64_t nce_pad(int64_t n) { return (n + 63) / 64 * 64; }

// ---- the all-f32 passes' work list ---------------------------------------------------------------------------------------
// The row counts n_k of a step (unique users / items of the batch: ~1880 and ~1620 of 2048 on the Yelp shape) exist only on
// the device, and the passes' work goes with n_k^2: a grid cut on the host for n_max runs 36 % more key blocks than the
// batch holds and leaves a sixth of the CUs without a workgroup.  So the f32 passes launch one workgroup per CU and every
// workgroup derives the SAME task list from the device counts: problem k is cut into qb_k = ceil(np_k / 128) query blocks
// (8 waves x 16 queries) times splits_k key ranges of per_k 32-key blocks, with the block budget T per task the one that
// minimises rounds(T) x T (rounds = tasks over workgroups; T >= kb_k / 16: the split partials' workspace holds 16).
struct NcePlan {
  int n[kNceMaxProblems], per[kNceMaxProblems], splits[kNceMaxProblems], first[kNceMaxProblems + 1];
};
constexpr int kNceQueryBlock = 128, kNceKeyBlock = 32, kNceMaxPer = 128;
constexpr int kNceReuseMax = 8192;      // pass 1 keeps its n x n weights for pass 2 up to this n (256 MB per problem)


__device__ __forceinline__ void nce_plan(const NceBatch& b, NcePlan& p) {
  int qb[kNceMaxProblems], kb[kNceMaxProblems];
  int t_min = 1;
  long long work = 0;
#pragma unroll
  for (int k = 0; k < kNceMaxProblems; ++k) {
    int n = 0;
    if (k < b.count) n = max(0, b.w[k].d_n ? min(*b.w[k].d_n, b.w[k].n_max) : b.w[k].n_max);
    p.n[k] = n;
    const int np = (int)nce_pad(n);
    qb[k] = (np + kNceQueryBlock - 1) / kNceQueryBlock;
    kb[k] = np / kNceKeyBlock;
    t_min = max(t_min, (kb[k] + kNceSplits - 1) / kNceSplits);
    work += (long long)qb[k] * kb[k];
  }
  int t = max(t_min, (int)((work + b.slots - 1) / b.slots));
  int best_t = t;
  long long best_cost = -1;
  for (int it = 0; it < 32; ++it, ++t) {
    int tasks = 0;
#pragma unroll
    for (int k = 0; k < kNceMaxProblems; ++k) tasks += qb[k] * ((kb[k] + t - 1) / t);
    const int rounds = (tasks + b.slots - 1) / b.slots;
    const long long cost = (long long)rounds * t;
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_t = t; }
    if (rounds <= 1) break;
  }
  p.first[0] = 0;
#pragma unroll
  for (int k = 0; k < kNceMaxProblems; ++k) {
    const int sp = kb[k] > 0 ? (kb[k] + best_t - 1) / best_t : 0;
    p.splits[k] = sp;
    p.per[k] = sp > 0 ? (kb[k] + sp - 1) / sp : 0;       // the even cut of kb_k over its splits (<= best_t)
    p.first[k + 1] = p.first[k] + qb[k] * sp;
  }
}


inline NceWs carve_nce(void* ws, int64_t n_max, int d) {
  NceWs w;
  const int64_t np = nce_pad(n_max);
  float* p = reinterpret_cast<float*>(ws);
  w.np = np;
  w.v1n = p; p += np * d;
  w.v2n = p; p += np * d;
  w.opart = p; p += (int64_t)kNceSplits * np * d;
  w.opart2 = p; p += (int64_t)kNceSplits * np * d;
  w.norm1 = p; p += np;
  w.norm2 = p; p += np;
  w.lpart = p; p += (int64_t)kNceSplits * np;
  w.invl = p; p += np;
  w.ediag = p; p += np;
  w.losspart = reinterpret_cast<double*>(p);     // np doubles (np is a multiple of 64: 8-byte aligned)
  uint16_t* h = reinterpret_cast<uint16_t*>(reinterpret_cast<double*>(p) + np);
  for (int v = 0; v < 2; ++v) {
    w.kq_hi[v] = h; h += np * d;
    w.kq_lo[v] = h; h += np * d;
    w.vt_hi[v] = h; h += np * d;
    w.vt_mid[v] = h; h += np * d;
    w.vt_lo[v] = h; h += np * d;
  }
  w.ticket = reinterpret_cast<int32_t*>(h);
  // (np / 16 + 1 counters, then the plan record on the next 128-byte line: the trailing 512 bytes of srh_infonce_ws_bytes)
  w.plan = reinterpret_cast<NcePlan*>(reinterpret_cast<char*>(h) + ((4 * (np / 16 + 1) + 127) / 128) * 128);
  w.etile = np <= kNceReuseMax ? reinterpret_cast<float*>(reinterpret_cast<char*>(h) + 4 * (np / 16) + 512) : nullptr;
  return w;
}

// normalise (and gather) the rows of both views; rows >= n are zero-filled
template <int LPR>
__device__ __forceinline__ void nce_prep_body(const NceBatch& batch, const unsigned bx, const unsigned by,
                                              const unsigned bz) {
  constexpr int G = 64 / LPR;
  const NceWs& w = batch.w[bz];
  const float* V1 = w.src1;
  const float* V2 = w.src2;
  const int32_t* idx = w.idx;
  const int n_max = w.n_max;
  const int n = w.d_n ? min(*w.d_n, n_max) : n_max;
  const int lane = threadIdx.x & 63, g = lane / LPR, sub = lane % LPR;
  const int i = (int)((bx * 256u + threadIdx.x) >> 6) * G + g;
  const bool second = by == 1;
  if (bx == 0 && by == 0)
    for (int k = threadIdx.x; k <= (int)(w.np / 16); k += 256) w.ticket[k] = 0;
  // rows_finish's per-workgroup loss partials: "not reported yet" (a bit pattern no sum of finite or non-finite terms has)
  if (bx == 0 && by == 1)
    for (int k = threadIdx.x; k < (int)w.np; k += 256) reinterpret_cast<unsigned long long*>(w.losspart)[k] = kLossUnreported;
  // the persistent passes' task list, cut ONCE per step from the live row counts: the passes and the finish read the record
  // (in every workgroup of the passes the same cut cost 4.9 k cycles of scalar divisions and dependent loads -- a tenth of
  // the kernel; tools/nce_stamps.py)
  if (batch.slots > 0 && bx == 0 && by == 0 && bz == 0 && threadIdx.x == 0) {
    NcePlan plan;
    nce_plan(batch, plan);
    *batch.w[0].plan = plan;
  }
  const float* V = second ? V2 : V1;
  float* out = second ? w.v2n : w.v1n;
  float* nrm = second ? w.norm2 : w.norm1;
  const bool in_pad = i < (int)w.np;
  const bool valid = i < n;
  const int src = valid ? (idx ? idx[i] : i) : 0;
  float4 v = reinterpret_cast<const float4*>(V)[(size_t)src * LPR + sub];
  const float ss = group_sum<LPR>(f4_dot(v, v));
  const float norm = sqrtf(ss);
  const float den = fmaxf(norm, 1e-12f);           // F.normalize eps
  float4 o = make_float4(v.x / den, v.y / den, v.z / den, v.w / den);
  if (!valid) o = f4_zero();
  if (in_pad) {
    reinterpret_cast<float4*>(out)[(size_t)i * LPR + sub] = o;
    if (sub == 0) nrm[i] = valid ? norm : 0.f;
    // split images for the MFMA passes: K / Q operands of the similarity product as SCALED FP16 hi + lo (kNceKqScale x,
    // an exact power of two: hi = f16(32 x), lo = f16(32 x - hi) represents x to 2^-22 -- |x| <= 1, so hi <= 32 and lo
    // stays a normal f16 down to |x| ~ 1e-3, below which its absolute error is < 2^-25 / 32), V operand of the P.V
    // product as bf16 hi + lo (its partner, the weights, needs f32's exponent range)
    if (batch.f32_only) return;
    const int view = second ? 1 : 0;
    constexpr int D = LPR * 4;
    const float f[4] = {o.x, o.y, o.z, o.w};
    uint16_t hi[4], mid[4], lo[4], kh[4], kl[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const __bf16 bh = (__bf16)f[t];
      const float r1 = f[t] - (float)bh;
      const __bf16 bm = (__bf16)r1;
      const __bf16 bl = (__bf16)(r1 - (float)bm);
      hi[t] = __builtin_bit_cast(uint16_t, bh);
      mid[t] = __builtin_bit_cast(uint16_t, bm);
      lo[t] = __builtin_bit_cast(uint16_t, bl);
      const float fs = f[t] * kNceKqScale;
      const _Float16 qh = (_Float16)fs;
      const _Float16 ql = (_Float16)(fs - (float)qh);
      kh[t] = __builtin_bit_cast(uint16_t, qh);
      kl[t] = __builtin_bit_cast(uint16_t, ql);
    }
    // this lane holds columns 4*sub .. 4*sub+3 of row i
    constexpr int DG = D / 4, KSL = D / 32, NTL = D / 16;
    {
      const int col0 = 4 * sub;
      const int gk = col0 / DG, within = col0 % DG, sk = within / 8, e0 = within % 8;   // 4 consecutive e
      const size_t at = ((((size_t)(i >> 4) * KSL + sk) * 64) + 16 * gk + (i & 15)) * 8 + e0;
      *reinterpret_cast<uint2*>(w.kq_hi[view] + at) = make_uint2(kh[0] | ((uint32_t)kh[1] << 16), kh[2] | ((uint32_t)kh[3] << 16));
      *reinterpret_cast<uint2*>(w.kq_lo[view] + at) = make_uint2(kl[0] | ((uint32_t)kl[1] << 16), kl[2] | ((uint32_t)kl[3] << 16));
    }
    const int blk = i >> 5, k = i & 31;
    const int gv = (k & 15) >> 2, ev = 4 * (k >> 4) + (k & 3);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int col = 4 * sub + t, c16v = col / NTL, nt = col % NTL;
      const size_t at = ((((size_t)blk * NTL + nt) * 64) + 16 * gv + c16v) * 8 + ev;
      w.vt_hi[view][at] = hi[t];
      w.vt_mid[view][at] = mid[t];
      if (kNcePvTerms == 6) w.vt_lo[view][at] = lo[t];
    }
  }
}

template <int LPR>
__global__ __launch_bounds__(256) void nce_prep(NceBatch batch) { nce_prep_body<LPR>(batch, blockIdx.x, blockIdx.y, blockIdx.z); }

// Horizontal fusion for the engine's step: the BPR kernels and the InfoNCE prep / finish kernels are
// all O(batch) and latency-bound with a wave or two per SIMD, and BPR phase 1 / the InfoNCE prep depend
// only on the forward pass, BPR phase 2 / the InfoNCE finish only on their own earlier phases -- so each
// pair shares one launch (workgroups [0, n_bpr) take the BPR role) and overlaps instead of queueing.
template <int LPR>
__global__ __launch_bounds__(256) void nce_prep_bpr1(NceBatch batch, BprArgs bpr, int n_bpr, int gpx) {
  if ((int)blockIdx.x < n_bpr) { bpr_phase1_body<LPR>(bpr, blockIdx.x); return; }
  const unsigned k = blockIdx.x - n_bpr;
  nce_prep_body<LPR>(batch, k % gpx, (k / gpx) & 1u, k / (2u * gpx));
}


// Split 16-bit arithmetic of nce_tile_lds (the default path).  An f32 operand x is carried as a short sum of 16-bit
// pieces and a product a.b as the cross terms that matter, each one MFMA at 16x the f32-MFMA rate, f32 accumulation:
//   similarity product S = Q K^T (the logits, and through them the loss): f16 pieces of 32 x -- hi = f16(32 x),
//     lo = f16(32 x - hi): 11 + 11 mantissa bits, x to 2^-22 -- and the three terms hi.hi + hi.lo + lo.hi on
//     v_mfma_f32_16x16x32_f16.  The rows are unit vectors, so the fixed scale keeps every piece in f16's normal range;
//     against fp64 the logits are off by 7e-8 (an f32 dot product: 3e-7), the loss by parts in 1e-12 (emulation of this
//     arithmetic with an exact accumulator: DESIGN.md 4.2; the MFMA's f32 accumulation adds the usual 1e-7).
//   P.V product (softmax weights times value rows -> the gradients): the weights span f32's exponent range (e^(-2/tau)
//     ... 1, and 1 / l(key) on top in pass 2), so they stay bf16 pieces -- hi, mid = bf16(x - hi) [, lo = bf16(x - hi -
//     mid)] on v_mfma_f32_16x16x32_bf16: three terms (2^-18 per product, gradients to 7e-7 relative) or, built with
//     SRH_NCE_PV_TERMS = 6, six (2^-27: gradients to 2e-9 of the exact product, below the f32 accumulator's own noise).
// Same dataflow as nce_tile: swapped S^T = K Q^T so a lane's 8 weights of a 32-key block ARE its A-operand fragment of
// the P.V product, whose V operand is read from the key-blocked transposed image (one 16-byte load per n-tile).  The
// summation index of every product may be permuted freely as long as both operands use the same permutation.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 ld_bf16x8(const uint16_t* p) {
  return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(p));
}


struct NceFinishArgs {
  float inv_tau, loss_scale;
  double* loss;
};

__device__ __forceinline__ void store_f64_sc1(double* p, double v) {
  asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

template <int LPR>
__device__ __forceinline__ float4 nce_norm_backward(float4 self, float4 dn, float norm) {
  const float proj = group_sum<LPR>(f4_dot(self, dn));
  if (norm > 1e-12f)
    return make_float4((dn.x - self.x * proj) / norm, (dn.y - self.y * proj) / norm,
                       (dn.z - self.z * proj) / norm, (dn.w - self.w * proj) / norm);
  return f4_scale(dn, 1e12f);
}


// Finish of one row (the LPR lanes of a row-group): fold both passes' split partials and form the
// gradients of both views through the normalisation (dv1, dv2: w.r.t. row idx[i] of the two source tables).  Returns the
// row's loss term (lse - s_ii) in the group's lane 0 (0 elsewhere / for padding rows).
struct NoHook { __device__ __forceinline__ void operator()() const {} };
// after_loads(): called once, after the row's loads have been issued and before the first of them is waited for (the caller's
// own next round trip goes out under this one's arithmetic)
template <int LPR, class Hook = NoHook>
__device__ __forceinline__ double nce_finish_row_grads(const NceWs& w, const NceFinishArgs& a, int n, int i, int sub,
                                                       int splits, float4& dv1, float4& dv2, Hook after_loads = Hook{}) {
  const bool valid = i < n;
  const int ii = valid ? i : 0;
  const size_t at = (size_t)ii * LPR + sub;
  // every load of the row is issued before the first use: this kernel is a chain of dependent round trips for
  // O(batch) bytes, and a split loop with a run-time trip count serialised eight of them
  const float4 va = reinterpret_cast<const float4*>(w.v1n)[at];
  const float4 vb = reinterpret_cast<const float4*>(w.v2n)[at];
  const float n1 = w.norm1[ii], n2 = w.norm2[ii];
  float4 O1 = f4_zero(), O2 = f4_zero();
  float l = 0.f;
  constexpr int kW = 10;                                // splits whose loads are in flight at a time: the persistent passes
                                                        // cut 8 .. 10 key ranges at the benchmark's shapes -- one round trip
  for (int k0 = 0; k0 < splits; k0 += kW) {
    float4 p1[kW], p2[kW];
    float lp[kW];
#pragma unroll
    for (int k = 0; k < kW; ++k) {
      p1[k] = p2[k] = f4_zero();
      lp[k] = 0.f;
      if (k0 + k < splits) {
        p1[k] = reinterpret_cast<const float4*>(w.opart + (size_t)(k0 + k) * w.np * (LPR * 4))[at];
        p2[k] = reinterpret_cast<const float4*>(w.opart2 + (size_t)(k0 + k) * w.np * (LPR * 4))[at];
        lp[k] = w.lpart[(size_t)(k0 + k) * w.np + ii];
      }
    }
    if (k0 == 0) after_loads();
#pragma unroll
    for (int k = 0; k < kW; ++k) {                     // split order: the order pass 2 folded 1 / l in
      O1 = f4_add(O1, p1[k]);
      O2 = f4_add(O2, p2[k]);
      l += lp[k];
    }
  }
  // l, O1, O2 are OFF-DIAGONAL sums (nce_tile_lds keeps the pair's own weight out); the diagonal enters here, exactly:
  //   e_ii = exp(s_ii / tau - 1 / tau) from the f32 dot product of the two rows,  L = l + e_ii,
  //   loss_i = log(L) - (s_ii / tau - 1 / tau) = log1p(l / e_ii),
  //   d/dn1_i = (O1 + e_ii v2_i) / L - v2_i = (O1 - l v2_i) / L,     d/dn2_i = O2 + (e_ii / L) v1_i - v1_i = O2 - (l / L) v1_i
  // -- no difference of nearly equal numbers anywhere, however sharp the softmax.
  if (splits <= 0) after_loads();
  const float coef = a.loss_scale * a.inv_tau / (float)n;
  const float sii = group_sum<LPR>(f4_dot(va, vb)) * a.inv_tau;
  const float eii = expf(sii - a.inv_tau);
  const float il = 1.0f / (l + eii);
  const float off = l * il;                      // 1 - p_ii
  const float4 dn1 = make_float4(coef * (O1.x * il - off * vb.x), coef * (O1.y * il - off * vb.y),
                                 coef * (O1.z * il - off * vb.z), coef * (O1.w * il - off * vb.w));
  const float4 dn2 = make_float4(coef * (O2.x - off * va.x), coef * (O2.y - off * va.y), coef * (O2.z - off * va.z),
                                 coef * (O2.w - off * va.w));
  dv1 = nce_norm_backward<LPR>(va, dn1, n1);
  dv2 = nce_norm_backward<LPR>(vb, dn2, n2);
  // log(L) - (s_ii / tau - 1 / tau).  log1p(l / e_ii) is the cancellation-free form of it, but e_ii underflows once
  // (1 - cos_ii) / tau > ~87 (below the tau >= 0.03 this entry accepts today: 2 / 0.03 = 67; kept for when that bound moves): there log(l) - (s_ii - 1 / tau) is exact to
  // rounding (e_ii no longer reaches l's last bit) and stays finite, like the reference's log_softmax
  const float li = (eii > 0.f && eii >= 1e-30f * l) ? log1pf(l / eii) : (logf(l) - (sii - a.inv_tau));
  return (valid && sub == 0) ? (double)li : 0.0;
}

// ... and the scatter of the two gradients with float atomics (the entry points without row -> slot lists)
template <int LPR>
__device__ __forceinline__ double nce_finish_row(const NceWs& w, const NceFinishArgs& a, int n, int i, int sub, float4* scr,
                                                 int splits) {
  const bool valid = i < n;
  const int dst = w.idx ? w.idx[valid ? i : 0] : (valid ? i : 0);
  float4 dv1, dv2;
  const double li = nce_finish_row_grads<LPR>(w, a, n, i, sub, splits, dv1, dv2);
  if (valid) {
    // atomic: BPR phase 2 shares this launch and adds to the same rows of g1.  g2 is a plain read-add-store
    // when the caller declares its rows exclusive (srh_infonce_problem_t::g2_exclusive)
    atomic_add_row<LPR>(w.g1 + (size_t)dst * LPR * 4, dv1, sub, scr);
    if (w.g2_plain) {
      float4* p2 = reinterpret_cast<float4*>(w.g2) + (size_t)dst * LPR + sub;
      *p2 = f4_add(*p2, dv2);
    } else {
      atomic_add_row<LPR>(w.g2 + (size_t)dst * LPR * 4, dv2, sub, scr);
    }
  }
  return li;
}

// LDS-staged form of nce_tile_bf16 (the default).  The register version is latency-bound: a workgroup's
// waves walk their key range 32 keys at a time and every step waits out an L2 round trip for 16 KB of
// operands with one wave per SIMD to hide it.  Here the workgroup first copies its WHOLE key range --
// the four fragment-linear images, <= 4 x 32 KB -- into LDS with direct global->LDS loads (the images
// are lane-linear 1 KB fragments, exactly the layout global_load_lds writes), pays the latency once,
// and then runs every step out of LDS with conflict-free ds_read_b128; WAVES = 8 waves (two per SIMD)
// share the stage so one wave's exp/convert VALU work overlaps the other's MFMAs.  PASS2 also folds the
// softmax denominators of its keys from pass 1's split partials while the copy is in flight, which
// retires the separate finish launch between the passes.  (Finishing a query block inside PASS2 by its
// last-arriving split was measured and lost: 37 us against 12.6 + 13.2 us -- write-through partials.)
__device__ __forceinline__ void glds16(const void* gsrc, void* ldst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)ldst, 16, 0, 0);
}
__device__ __forceinline__ bf16x8 lds_bf16x8(const unsigned char* p) {
  return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(p));
}
__device__ __forceinline__ f16x8 lds_f16x8(const unsigned char* p) {
  return __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(p));
}
__device__ __forceinline__ f16x8 ld_f16x8(const uint16_t* p) {
  return __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(p));
}

// keys per LDS stage: 256 at d = 64 with 4 images (128 KB), half of that when the stage carries a fifth image (6-term
// P.V) or a workgroup's whole key range is 128 keys anyway (16 splits of 2048)
template <int D, int PVT>
constexpr int nce_chunk() { return (PVT == 6 || kNceUsedSplits == 16 ? 8192 : 16384) / D; }
template <int D, int PVT>
constexpr int nce_lds_bytes() { return (PVT == 6 ? 5 : 4) * nce_chunk<D, PVT>() * D * 2 + nce_chunk<D, PVT>() * 4; }

template <int D, bool PASS2, int QT, int WAVES, int PVT>
__global__ __launch_bounds__(64 * WAVES) void nce_tile_lds(NceBatch batch, float inv_tau) {
  constexpr int NT = D / 16, KS = D / 32;
  constexpr int NIMG = PVT == 6 ? 5 : 4;     // kq_hi, kq_lo, vt_hi, vt_mid [, vt_lo]
  constexpr int CHUNK = nce_chunk<D, PVT>();
  constexpr int SPAN = CHUNK * D * 2;        // bytes of one image of one stage
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* invl_s = reinterpret_cast<float*>(smem + NIMG * SPAN);
  const NceWs& w = batch.w[blockIdx.z];
  const int n = w.d_n ? min(*w.d_n, w.n_max) : w.n_max;
  const int np = (int)w.np;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c16 = lane & 15, g = lane >> 4;
  const int q0 = (blockIdx.x * WAVES + wv) * (16 * QT);
  const int ks = blockIdx.y;
  if (blockIdx.x * (16 * QT * WAVES) >= np) return;              // whole workgroup beyond this problem's rows
  const int qv = PASS2 ? 1 : 0, kv = PASS2 ? 0 : 1;
  const int per = ((np + batch.splits - 1) / batch.splits + 31) / 32 * 32;
  const int kb = ks * per, ke = min(np, kb + per);
  const bool wave_live = q0 < np;

  f16x8 qh[QT][KS], ql[QT][KS];
#pragma unroll
  for (int t = 0; t < QT; ++t)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int tile = min((q0 >> 4) + t, (np >> 4) - 1);
      const size_t at = (((size_t)tile * KS + s) * 64 + lane) * 8;
      qh[t][s] = ld_f16x8(w.kq_hi[qv] + at);
      ql[t][s] = ld_f16x8(w.kq_lo[qv] + at);
    }
  const float s_scale = inv_tau * kNceInvKqScale2;       // the images hold 32 x: products come out 1024 x (exact)
  floatx4 O[QT][NT];
#pragma unroll
  for (int t = 0; t < QT; ++t)
#pragma unroll
    for (int u = 0; u < NT; ++u) O[t][u] = (floatx4){0.f, 0.f, 0.f, 0.f};
  float lsum[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) lsum[t] = 0.f;

  for (int c0 = kb; c0 < ke; c0 += CHUNK) {
    const int cn = min(CHUNK, ke - c0);                  // a multiple of 32
    if (c0 != kb) __syncthreads();                       // the previous stage has been read by every wave
    const int n_kb = cn * D * 2 / 1024;                  // 1 KB fragments per image in this stage
    const unsigned char* src[5] = {
        reinterpret_cast<const unsigned char*>(w.kq_hi[kv]) + (size_t)(c0 >> 4) * KS * 1024,
        reinterpret_cast<const unsigned char*>(w.kq_lo[kv]) + (size_t)(c0 >> 4) * KS * 1024,
        reinterpret_cast<const unsigned char*>(w.vt_hi[kv]) + (size_t)(c0 >> 5) * NT * 1024,
        reinterpret_cast<const unsigned char*>(w.vt_mid[kv]) + (size_t)(c0 >> 5) * NT * 1024,
        reinterpret_cast<const unsigned char*>(w.vt_lo[kv]) + (size_t)(c0 >> 5) * NT * 1024};
#pragma unroll
    for (int img = 0; img < NIMG; ++img)
      for (int k = wv; k < n_kb; k += WAVES)
        glds16(src[img] + (size_t)k * 1024 + lane * 16, smem + img * SPAN + k * 1024);
    if (PASS2) {
      // 1 / l(key) from pass 1's split partials, in split order (the order nce_finish uses)
      for (int t = threadIdx.x; t < cn; t += 64 * WAVES) {
        float l = 0.f;
        for (int sp = 0; sp < batch.splits; ++sp) l += w.lpart[(size_t)sp * np + c0 + t];
        invl_s[t] = 1.0f / (l + w.ediag[c0 + t]);       // (lpart holds the off-diagonal sums; rows >= n: unused)
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!wave_live) continue;

    // one 32-key block: its operand fragments out of the LDS stage ...
    struct Blk { f16x8 kh[2][KS], kl[2][KS]; bf16x8 vh[NT], vm[NT], vl[PVT == 6 ? NT : 1]; float il[2][4]; };
    auto load_blk = [&](int j0, Blk& b) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const int off = (((j0 >> 4) + h) * KS + s) * 1024 + lane * 16;
          b.kh[h][s] = lds_f16x8(smem + off);
          b.kl[h][s] = lds_f16x8(smem + SPAN + off);
        }
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const int off = ((j0 >> 5) * NT + u) * 1024 + lane * 16;
        b.vh[u] = lds_bf16x8(smem + 2 * SPAN + off);
        b.vm[u] = lds_bf16x8(smem + 3 * SPAN + off);
        if (PVT == 6) b.vl[u] = lds_bf16x8(smem + 4 * SPAN + off);
      }
      if (PASS2) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int r = 0; r < 4; ++r) b.il[h][r] = invl_s[j0 + 16 * h + 4 * g + r];
      }
    };
    // ... and the block's work: logits (three f16 MFMA terms), weights, P.V (three bf16 terms)
    auto compute_blk = [&](int j0, const Blk& b) {
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        floatx4 a[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          a[h] = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            // smallest terms first into the f32 accumulator
            a[h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b.kl[h][s], qh[t][s], a[h], 0, 0, 0);
            a[h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b.kh[h][s], ql[t][s], a[h], 0, 0, 0);
            a[h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b.kh[h][s], qh[t][s], a[h], 0, 0, 0);
          }
        }
        bf16x8 ph, pm, pl;
        const int qrow = q0 + 16 * t + c16;              // the query whose weights this lane holds
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = c0 + j0 + 16 * h + 4 * g + r;
            float e = __expf(fmaf(a[h][r], s_scale, -inv_tau));
            // The pair's OWN weight (key == query: the positive of InfoNCE) stays out of the accumulated sums.  Its part
            // of the gradient, p_ii v_i - v_i, is a cancellation -- the closer p_ii is to 1 (small tau, a trained model)
            // the more of the sum's rounding error it exposes (measured with it inside the sums: gradient errors of
            // 3e-3 ... 8e-2 at tau = 0.05 on correlated views, all-f32 path included) -- so the finish kernel forms
            // -(1 - p_ii) v_i = -(l' / l) v_i from the off-diagonal sum l' directly, and the loss as log1p(l' / e_ii).
            const bool own = key == qrow;
            if (!PASS2 && own && key < n) w.ediag[qrow] = e;
            if (PASS2) e *= b.il[h][r];
            const float wt = (key < n && !own) ? e : 0.f;
            lsum[t] += wt;
            const __bf16 bh = (__bf16)wt;
            const float r1 = wt - (float)bh;
            const __bf16 bm = (__bf16)r1;
            ph[4 * h + r] = bh;
            pm[4 * h + r] = bm;
            if (PVT == 6) pl[4 * h + r] = (__bf16)(r1 - (float)bm);
          }
#pragma unroll
        for (int u = 0; u < NT; ++u) {
          // the block's partial product summed smallest terms first, then added to the running output in one MFMA chain
          if (PVT == 6) {
            O[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pl, b.vh[u], O[t][u], 0, 0, 0);
            O[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, b.vl[u], O[t][u], 0, 0, 0);
            O[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pm, b.vm[u], O[t][u], 0, 0, 0);
          }
          O[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pm, b.vh[u], O[t][u], 0, 0, 0);
          O[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, b.vm[u], O[t][u], 0, 0, 0);
          O[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, b.vh[u], O[t][u], 0, 0, 0);
        }
      }
    };
    // (Round 4, measured and dropped: two register sets in ping-pong -- the fragments of block j + 1 read under the MFMAs of
    // block j, 228 VGPRs -- left the step where it was, 0.2788 against 0.2770 / 0.2784 ms for the product before / after in
    // the same session: the exposed LDS round trip per block is not what these passes wait for.)
    for (int j0 = 0; j0 < cn; j0 += 32) {
      Blk b;
      load_blk(j0, b);
      compute_blk(j0, b);
    }
  }
  if (!wave_live) return;

  float* obase = PASS2 ? w.opart2 : w.opart;
  if (wave_live) {
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      const int qt0 = q0 + 16 * t;
      if (qt0 >= np) break;
      float* op = obase + ((size_t)ks * np + qt0) * D;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* rowp = op + (size_t)(4 * g + r) * D + c16 * NT;
#pragma unroll
        for (int u = 0; u < NT / 4; ++u) {
          const float4 v = make_float4(O[t][4 * u + 0][r], O[t][4 * u + 1][r], O[t][4 * u + 2][r], O[t][4 * u + 3][r]);
          st_f4<kNceWT>(reinterpret_cast<float4*>(rowp) + u, v);
        }
      }
      if (!PASS2) {
        float l = lsum[t];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        if (g == 0) w.lpart[(size_t)ks * np + qt0 + c16] = l;
      }
    }
  }
}


// ---- SRH_NCE_F32: both n x n x d products of a pass on v_mfma_f32_16x16x4_f32 (exact f32 multiply-adds) -----------------
// Same dataflow and conventions as nce_tile_lds (swapped S^T = K Q^T so a lane's weights ARE its A fragment of the P.V
// product; the pair's own weight kept out of the sums, e_ii left in w.ediag), on the f32 rows v1n / v2n themselves.
// The matrix pipe is the bound (2 x 2 n^2 d flops per pass at 64 flop / clk / SIMD), so the kernel is built to keep it
// issuing:
//   * one PERSISTENT workgroup per CU, 8 waves x 16 queries, tasks cut on the device from the live row counts (nce_plan);
//   * the task's key rows arrive through a ring of eight 32-key blocks in LDS, filled by direct global->LDS loads two
//     chunks (4 blocks) ahead -- one barrier per 64 keys; ONE f32 image serves both products: rows are stored with their
//     16-byte chunks XOR-swizzled by the row number, the S product reads chunk g + 4 t of row c16 and the P.V product chunk
//     (c16 + 4) % 16 of row 4 g + r: both conflict-free under ds_read_b128's lane groups {0-3, 12-15, 20-27}, ...;
//   * a wave's own stream is software-pipelined: S(j + 1)'s 32 MFMAs are issued with block j's exp / mask VALU work in
//     their shadow, then P.V(j)'s 32 MFMAs with the LDS reads of block j + 2's operands in theirs.
template <int D>
struct NceF32 {
  static constexpr int kBlockBytes = kNceKeyBlock * D * 4;       // one 32-key block of f32 rows
  static constexpr int kSlots = 8;
  static constexpr int kLoadsPerBlock = kBlockBytes / 1024 / 8;  // direct-load instructions per wave per block
  static constexpr int kLds = kSlots * kBlockBytes + kNceMaxPer * kNceKeyBlock * 4;
};

// LDS reads the compiler does not see as LDS reads.  Every ds_read the compiler emits itself after a direct global->LDS
// load waits for vmcnt(0) -- it cannot tell which bytes the load fills -- which would drain the ring's run-ahead at every
// block.  These are issued as opaque instructions; lds_wait() is the lgkmcnt(0) their consumers sit behind (the operands
// pass through it, so no use can be scheduled above it).
template <int OFFSET>
__device__ __forceinline__ floatx4 lds_read_f4_async(unsigned addr) {
  static_assert(OFFSET >= 0 && OFFSET < 65536 && OFFSET % 16 == 0, "ds_read_b128 offset field");
  floatx4 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFFSET));
  return r;
}
template <int N>
__device__ __forceinline__ void lds_wait(floatx4 (&x)[N]);
// two operand sets behind one wait (the wait covers every outstanding read, but only the registers it NAMES are ordered
// behind it for the compiler: a set left out may be consumed above it)
__device__ __forceinline__ void lds_wait(floatx4 (&x)[4], floatx4 (&y)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]));
}
template <int N>
__device__ __forceinline__ void lds_wait(floatx4 (&x)[N]) {
  static_assert(N == 4 || N == 8 || N == 16, "lds_wait: 4, 8 or 16 registers quads");
  if constexpr (N == 4) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
  } else if constexpr (N == 8) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
  } else {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),
                   "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]));
  }
}

// 16 bytes per lane from `base` + lane_off + wave_off (raw buffer of `bytes` bytes) into LDS at `lds` + 16 lane: one
// buffer_load_dwordx4 ... lds.  (A __device__ function of its own: with the buffer-resource type in a __global__ body the
// host pass of hipcc 7.2 silently drops the kernel's stub.)
__device__ __forceinline__ void buffer_to_lds16(const void* base, int bytes, void* lds, int lane_off, int wave_off) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, lane_off, wave_off, 0, 0);
}

#ifdef SRH_NCEF32_STAMPS
// (laboratory builds only: tools/spmm_lab/build_alt.sh stamps "-DSRH_NCEF32_STAMPS"; per wave {start, planned, first chunk in,
//  key loop done, stores done, blocks, task} on the shader clock -- read by tools/nce_stamps.py through srh_debug_nce_stamps)
__device__ unsigned long long g_nce_stamps[2][512 * 8 * 8];
#define SRH_STAMP(k) do { if (lane == 0) g_nce_stamps[PASS2 ? 1 : 0][((size_t)blockIdx.x * 8 + wv) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#define SRH_STAMP_VAL(k, v) do { if (lane == 0) g_nce_stamps[PASS2 ? 1 : 0][((size_t)blockIdx.x * 8 + wv) * 8 + (k)] = (unsigned long long)(v); } while (0)
#else
#define SRH_STAMP(k) do {} while (0)
#define SRH_STAMP_VAL(k, v) do {} while (0)
#endif

// REUSE (every problem of the batch has its n x n weight array: n <= kNceReuseMax): pass 1 also stores the weight of every
// (query i, key j) pair -- masked: 0 for the pair's own, for keys >= n and for queries >= n -- transposed, so that the lane
// holding query j of pass 2 finds its four keys 4 g .. 4 g + 3 in ONE 16-byte load; pass 2 then runs no S product, no exp
// and no masks: half its MFMAs.  (The f32 MFMA and the VALU do not overlap on this chip, so what a pass costs is the SUM of
// the two: 64 MFMAs x 32 cycles per block and wave against 32 in pass 2 now.)
__device__ __forceinline__ void st_f1_wt(float* p, float v) {
  asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

template <int D, bool PASS2, bool REUSE>
__global__ __launch_bounds__(512) void nce_tile_f32(NceBatch batch, float inv_tau) {
  constexpr int DQ = D / 4;            // k-steps of the S product (dims per lane)
  constexpr int NT = D / 16;           // 16-column n-tiles of the P.V product
  constexpr int NV = NT / 4;           // 16-byte chunks of a row a lane reads for the P.V product
  constexpr int KQ = DQ / 4;           // 16-byte chunks of a row a lane reads for the S product
  constexpr int RB = D * 4;            // bytes per row
  using Cfg = NceF32<D>;
  constexpr int LPB = Cfg::kLoadsPerBlock;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* invl_s = reinterpret_cast<float*>(smem + Cfg::kSlots * Cfg::kBlockBytes);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  SRH_STAMP(0);
  const NcePlan plan = *batch.w[0].plan;
  SRH_STAMP(1);
  const int c16 = lane & 15, g = lane >> 4;
  const int cidx = (c16 + 4) & 15;

  // per-lane LDS byte offsets inside a block (the swizzle: chunk c of row r sits at chunk (c & ~15) | ((c ^ r) & 15));
  // the tile half h and the chunk's high bit are immediates of the reads
  unsigned off_s[4], off_v[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) off_s[t] = lds0 + c16 * RB + (((g + 4 * t) ^ c16) & 15) * 16;
#pragma unroll
  for (int r = 0; r < 4; ++r) off_v[r] = lds0 + (4 * g + r) * RB + ((cidx ^ (4 * g + r)) & 15) * 16;
  // the direct loads of one block: instruction i of this wave fills LDS bytes [1024 (LPB wv + i), +1024) of the slot
  int ld_off[LPB];
#pragma unroll
  for (int i = 0; i < LPB; ++i) {
    const int pos = (LPB * wv + i) * 1024 + lane * 16;
    const int row = pos / RB, phys = (pos % RB) >> 4;
    ld_off[i] = row * RB + ((phys & ~15) | ((phys ^ row) & 15)) * 16;
  }

  const int n_tasks = plan.first[kNceMaxProblems];
  for (int task = blockIdx.x; task < n_tasks; task += gridDim.x) {
    int k = 0, first = 0, per = plan.per[0], splits = plan.splits[0], n = plan.n[0];
#pragma unroll
    for (int q = 1; q < kNceMaxProblems; ++q)
      if (task >= plan.first[q]) { k = q; first = plan.first[q]; per = plan.per[q]; splits = plan.splits[q]; n = plan.n[q]; }
    const NceWs& w = batch.w[k];
    const int np = (int)nce_pad(n);
    const size_t stride = (size_t)w.np;
    const float* lpart = w.lpart;
    float* ediag = w.ediag;
    const int local = task - first;
    const int qblk = local / splits, split = local % splits;
    const int b0 = split * per;
    const int nblk = min(per, np / kNceKeyBlock - b0);     // >= 1
    const int q0 = qblk * kNceQueryBlock + wv * 16;
    const bool wave_live = q0 < np;
    const float* Qv = PASS2 ? w.v2n : w.v1n;
    const unsigned char* Kv = reinterpret_cast<const unsigned char*>(PASS2 ? w.v1n : w.v2n) + (size_t)b0 * Cfg::kBlockBytes;
    float* opart = (PASS2 ? w.opart2 : w.opart) + ((size_t)split * stride + q0) * D;
    // pass 1 writes etile[key][query q0 + c16]; pass 2 (queries = pass 1's keys) reads etile[query q0 + c16][key .. key + 3]
    float* et = REUSE ? (PASS2 ? w.etile + (size_t)min(q0 + c16, np - 1) * stride + (size_t)b0 * kNceKeyBlock + 4 * g
                               : w.etile + ((size_t)b0 * kNceKeyBlock + 4 * g) * stride + q0 + c16)
                      : nullptr;
    if (task != (int)blockIdx.x) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the previous task's ring and 1 / l are read out

    // (buffer_load ... lds, not global_load_lds: the compiler files the FLAT-encoded form under "may return out of
    // order" and turns every vmcnt it inserts itself into vmcnt(0) while one is in flight -- the ring's run-ahead gone)
    const int k_bytes = nblk * Cfg::kBlockBytes;
    auto issue_chunk = [&](int c) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int b = min(2 * c + jj, nblk - 1);                    // (past the range: a harmless re-load, never read)
        unsigned char* dst = smem + ((2 * c + jj) & (Cfg::kSlots - 1)) * Cfg::kBlockBytes + LPB * wv * 1024;
#pragma unroll
        for (int i = 0; i < LPB; ++i)
          buffer_to_lds16(Kv, k_bytes, dst + i * 1024, ld_off[i], b * Cfg::kBlockBytes);
      }
    };
    // the ring's first chunk first: the longest latency of the prologue; the query rows and (pass 2) the 1 / l fold go under
    // it, the run-ahead chunks after them (so that "all but the 4 LPB youngest" below names exactly chunk 0 and these)
    issue_chunk(0);
    constexpr bool kLogits = !(PASS2 && REUSE);                     // this pass forms the logits itself
    float qreg[DQ];
    if constexpr (kLogits) {
      const float* src = Qv + (size_t)min(q0 + c16, np - 1) * D + 4 * g;
#pragma unroll
      for (int t = 0; t < KQ; ++t) {
        const float4 x = *reinterpret_cast<const float4*>(src + 16 * t);
        qreg[4 * t + 0] = x.x; qreg[4 * t + 1] = x.y; qreg[4 * t + 2] = x.z; qreg[4 * t + 3] = x.w;
      }
    }
    if (PASS2) {
      // 1 / l(key) from pass 1's split partials, in split order (lpart holds the off-diagonal sums; rows >= n: unused)
      for (int t = threadIdx.x; t < nblk * kNceKeyBlock; t += 512) {
        const int key = b0 * kNceKeyBlock + t;
        float lp[kNceSplits];
#pragma unroll
        for (int sp = 0; sp < kNceSplits; ++sp) lp[sp] = sp < splits ? lpart[(size_t)sp * stride + key] : 0.f;   // all in flight
        const float ed = ediag[key];
        float l = 0.f;
#pragma unroll
        for (int sp = 0; sp < kNceSplits; ++sp) l += lp[sp];       // split order (adding 0 for the splits that do not exist)
        invl_s[t] = 1.0f / (l + ed);
      }
    }
    issue_chunk(1);
    issue_chunk(2);
    // RESIDENT: the task's whole key range fits the ring (<= 8 blocks: every task of a batch of n <= 4096 rows) -- all of it
    // is requested now and the key loop runs without a barrier.  (Each chunk boundary of the ring costs ~1.3 k cycles with
    // the matrix pipe idle: all eight waves meet at the barrier, issue their two direct loads -- 60-185 cycles each -- and
    // wait out an LDS round trip at the same time; a 7-block task has three of them.)
    const bool resident = nblk <= Cfg::kSlots;
    if (resident) {
      issue_chunk(3);
      // chunk 0 has landed once all but the 6 LPB youngest loads are back (lgkmcnt: the 1 / l stores)
      if constexpr (LPB == 1) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
      // ... all but the 4 LPB youngest
      if constexpr (LPB == 1) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

    // S operands of one block: K rows 16 h + c16, chunks g + 4 t   (kq[KQ h + t], element e <-> k-step 4 t + e)
    floatx4 kq[2 * KQ];
    // (address = one add per operand set and slot: the tile half and the chunk's high bit ride in the offset field --
    //  the f32 MFMA and the vector ALU do not overlap on this chip, tools/microbench/mfma_f32_valu.hip: every VALU
    //  instruction of the key loop is paid in full)
    auto issue_k = [&](int j) {
      const unsigned base = (unsigned)((j & (Cfg::kSlots - 1)) * Cfg::kBlockBytes);
      unsigned a[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) a[t] = off_s[t] + base;
      kq[0] = lds_read_f4_async<0>(a[0]); kq[1] = lds_read_f4_async<0>(a[1]);
      kq[2] = lds_read_f4_async<0>(a[2]); kq[3] = lds_read_f4_async<0>(a[3]);
      if constexpr (KQ == 8) {
        kq[4] = lds_read_f4_async<256>(a[0]); kq[5] = lds_read_f4_async<256>(a[1]);
        kq[6] = lds_read_f4_async<256>(a[2]); kq[7] = lds_read_f4_async<256>(a[3]);
      }
      kq[KQ + 0] = lds_read_f4_async<16 * RB>(a[0]); kq[KQ + 1] = lds_read_f4_async<16 * RB>(a[1]);
      kq[KQ + 2] = lds_read_f4_async<16 * RB>(a[2]); kq[KQ + 3] = lds_read_f4_async<16 * RB>(a[3]);
      if constexpr (KQ == 8) {
        kq[KQ + 4] = lds_read_f4_async<16 * RB + 256>(a[0]); kq[KQ + 5] = lds_read_f4_async<16 * RB + 256>(a[1]);
        kq[KQ + 6] = lds_read_f4_async<16 * RB + 256>(a[2]); kq[KQ + 7] = lds_read_f4_async<16 * RB + 256>(a[3]);
      }
    };
    // P.V operands of half a block: V rows 16 h + 4 g + r, chunks cidx + 16 v   (vv[NV r + v], element u' <-> n-tile 4 v + u')
    auto issue_v = [&](int j, auto half_tag, floatx4 (&vv)[4 * NV]) {
      constexpr int H = decltype(half_tag)::value;
      const unsigned base = (unsigned)((j & (Cfg::kSlots - 1)) * Cfg::kBlockBytes);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned a = off_v[r] + base;
        vv[NV * r] = lds_read_f4_async<16 * H * RB>(a);
        if constexpr (NV == 2) vv[NV * r + 1] = lds_read_f4_async<16 * H * RB + 256>(a);
      }
    };
    floatx4 acc[2];
    auto s_product = [&]() {
      acc[0] = (floatx4){0.f, 0.f, 0.f, 0.f};
      acc[1] = acc[0];
#pragma unroll
      for (int s = 0; s < DQ; ++s) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(kq[s >> 2][s & 3], qreg[s], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(kq[KQ + (s >> 2)][s & 3], qreg[s], acc[1], 0, 0, 0);
      }
    };
    SRH_STAMP(2);
    SRH_STAMP_VAL(5, nblk);
    SRH_STAMP_VAL(6, task);
    floatx4 O[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) O[u] = (floatx4){0.f, 0.f, 0.f, 0.f};
    float lsum = 0.f;
    const int qrow = q0 + c16;                                      // the query whose weights this lane holds

    // one block: [P.V operands of half 0 | S(j + 1) beside block j's weights] then [P.V half 0 | operands of half 1]
    // then [P.V half 1 | S operands of block j + 2]
    using Half0 = std::integral_constant<int, 0>;
    using Half1 = std::integral_constant<int, 1>;
    // weight of a logit: exp(s / tau - 1 / tau) as one fma and one v_exp_f32 (2^x)
    const float ex_scale = inv_tau * 1.44269504088896340736f;
    auto block_step = [&](int j, auto more_tag, auto edge_tag) {
      constexpr bool kMore = decltype(more_tag)::value;             // block j + 1 exists: its S product runs here
      constexpr bool kEdge = decltype(edge_tag)::value;             // the block holds keys >= n or this wave's own pairs
      constexpr bool kHalves = D > 64;                              // d = 128: the second half's operands follow the first's MFMAs
      floatx4 v0[4 * NV], v1[4 * NV];
      const floatx4 a0 = acc[0], a1 = acc[1];
      issue_v(j, Half0{}, v0);
      if constexpr (!kHalves) issue_v(j, Half1{}, v1);
      if constexpr (kMore) s_product();
      float wgt[2][4];
      float own_e = 0.f;
      bool has_own = false;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kl = kNceKeyBlock * j + 16 * h + 4 * g + r;
          float e = __builtin_amdgcn_exp2f(fmaf(h == 0 ? a0[r] : a1[r], ex_scale, -ex_scale));
          if constexpr (kEdge) {
            const int key = kNceKeyBlock * b0 + kl;
            const bool own = key == qrow;
            if (!PASS2) { own_e = own ? e : own_e; has_own |= own; }
            if (PASS2) e *= invl_s[kl];
            e = (key < n && !own) ? e : 0.f;
          } else {
            if (PASS2) e *= invl_s[kl];
          }
          lsum += e;
          wgt[h][r] = e;
        }
      if constexpr (kEdge) {
        if (!PASS2 && has_own && qrow < n) ediag[qrow] = own_e;     // the pair's own weight: nce_finish_row folds it in
      }
      if constexpr (REUSE && !PASS2) {
        // the weights pass 2 will read back (write-through: another XCD reads them next, nobody here does)
        float* row = et + (size_t)(kNceKeyBlock * j) * stride;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int r = 0; r < 4; ++r) st_f1_wt(row + (size_t)(16 * h + r) * stride, wgt[h][r]);
      }
      if constexpr (kHalves) {
        lds_wait(v0);
        issue_v(j, Half1{}, v1);
      } else {
        lds_wait(v0, v1);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int u = 0; u < NT; ++u)
          O[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wgt[0][r], v0[NV * r + (u >> 2)][u & 3], O[u], 0, 0, 0);
      if constexpr (kHalves) lds_wait(v1);
      if constexpr (kMore) issue_k(j + 2);                          // (past the range: stale bytes, never used)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int u = 0; u < NT; ++u)
          O[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wgt[1][r], v1[NV * r + (u >> 2)][u & 3], O[u], 0, 0, 0);
      // the S operands are settled HERE, in the block that issued their reads, not at their use in the next iteration: a
      // value that crosses the loop edge may pass through compiler-made copies, and a copy of a register whose read is
      // still in flight copies stale bytes (tests/test_isa_async_lds.py checks the built kernel for exactly that)
      if constexpr (kMore) lds_wait(kq);
    };

    // pass 2 on pass 1's weights: [V operands out of LDS | weights of block j + 2 out of memory] then the block's 32 P.V
    // MFMAs.  The weights arrive masked (own pairs, keys >= n of pass 1 = queries here); what pass 2 still masks is ITS keys
    // >= n (pass 1's padding queries), in the one block that can hold them.
    floatx4 e_now[2], e_next[2], e_far[2];
    auto load_e = [&](int j, floatx4 (&e)[2]) {
      const int jj = min(j, nblk - 1);                              // (past the range: a re-load, never used)
      const float* src = et + (size_t)kNceKeyBlock * jj;
      e[0] = *reinterpret_cast<const floatx4*>(src);
      e[1] = *reinterpret_cast<const floatx4*>(src + 16);
    };
    auto reuse_step = [&](int j, auto edge_tag) {
      constexpr bool kEdge = decltype(edge_tag)::value;
      constexpr bool kHalves = D > 64;
      floatx4 v0[4 * NV], v1[4 * NV];
      issue_v(j, Half0{}, v0);
      if constexpr (!kHalves) issue_v(j, Half1{}, v1);
      load_e(j + 2, e_far);
      float wgt[2][4];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const floatx4 il = *reinterpret_cast<const floatx4*>(invl_s + kNceKeyBlock * j + 16 * h + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float wt = e_now[h][r] * il[r];
          if constexpr (kEdge) wt = (kNceKeyBlock * (b0 + j) + 16 * h + 4 * g + r < n) ? wt : 0.f;
          wgt[h][r] = wt;
        }
      }
      if constexpr (kHalves) {
        lds_wait(v0);
        issue_v(j, Half1{}, v1);
      } else {
        lds_wait(v0, v1);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int u = 0; u < NT; ++u)
          O[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wgt[0][r], v0[NV * r + (u >> 2)][u & 3], O[u], 0, 0, 0);
      if constexpr (kHalves) lds_wait(v1);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int u = 0; u < NT; ++u)
          O[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wgt[1][r], v1[NV * r + (u >> 2)][u & 3], O[u], 0, 0, 0);
      e_now[0] = e_next[0]; e_now[1] = e_next[1];
      e_next[0] = e_far[0]; e_next[1] = e_far[1];
    };

    if constexpr (kLogits) {
      if (wave_live) {
        issue_k(0);
        lds_wait(kq);
        s_product();                                                // (block 0's 32 MFMAs cover the wait below)
      }
      if (resident) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");      // the whole range is in, everywhere
      if (wave_live) {
        issue_k(1);                                                 // (chunk 0 = blocks 0 and 1 has landed; one block: unused)
        lds_wait(kq);
      }
    } else {
      if (wave_live) {
        load_e(0, e_now);
        load_e(1, e_next);
      }
      if (resident) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
    for (int j = 0; j < nblk; ++j) {
      if (!resident && (j & 1) == 0) {
        // chunks <= c + 1 have landed once only chunk c + 2's loads are out; then every wave is past chunk c - 1 and its
        // slots take chunk c + 3
        if constexpr (LPB == 1) asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
        issue_chunk((j >> 1) + 3);
      }
      if (!wave_live) continue;
      // masks only where they can bite: a block reaching past the live rows, or one that holds this wave's own pairs
      const int key0 = kNceKeyBlock * (b0 + j);
      if constexpr (!kLogits) {
        if (key0 + kNceKeyBlock > n) reuse_step(j, std::true_type{});
        else reuse_step(j, std::false_type{});
        continue;
      }
      const bool edge = key0 + kNceKeyBlock > n || (key0 < q0 + 16 && key0 + kNceKeyBlock > q0);
      if (j + 1 < nblk) {
        if (edge) block_step(j, std::true_type{}, std::true_type{});
        else block_step(j, std::true_type{}, std::false_type{});
      } else {
        if (edge) block_step(j, std::false_type{}, std::true_type{});
        else block_step(j, std::false_type{}, std::false_type{});
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // (the ring's run-ahead loads: nothing may land later)
    SRH_STAMP(3);
    if (!wave_live) continue;

    // O[4 v + u'][r] = out[query q0 + 4 g + r][column 4 (cidx + 16 v) + u']
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int v = 0; v < NV; ++v)
        st_f4<kNceWT>(reinterpret_cast<float4*>(opart + (size_t)(4 * g + r) * D + 4 * (cidx + 16 * v)),
                      make_float4(O[4 * v + 0][r], O[4 * v + 1][r], O[4 * v + 2][r], O[4 * v + 3][r]));
    if (!PASS2) {
      lsum += __shfl_xor(lsum, 16);
      lsum += __shfl_xor(lsum, 32);
      if (g == 0) const_cast<float*>(lpart)[(size_t)split * stride + qrow] = lsum;
    }
    SRH_STAMP(4);
  }
}

// One finish for both passes (used with nce_tile_lds): folds the split partials, turns them into the
// gradients of both views (through the normalisation) and scatters them; the loss partials are folded
// in workgroup order by whichever workgroup of the problem arrives last (write-through partial ->
// vmcnt(0) -> relaxed agent-scope ticket -> acquire), so the reported loss is bitwise reproducible.
template <int LPR>
__device__ __forceinline__ void nce_finish_both_body(const NceBatch& batch, const NceFinishArgs& a, const unsigned bx,
                                                     const unsigned bz) {
  constexpr int G = 64 / LPR;
  const NceWs& w = batch.w[bz];
  const int n = w.d_n ? min(*w.d_n, w.n_max) : w.n_max;
  const int wave = (int)((bx * 256u + threadIdx.x) >> 6);
  const int n_waves = (int)(w.np / G);
  if (n <= 0 || wave >= n_waves) return;
  const int lane = threadIdx.x & 63, g = lane / LPR, sub = lane % LPR;
  __shared__ float4 s_scr[4][64];
  int splits = batch.splits;
  if (batch.slots > 0) {            // the persistent passes' cut of this problem (nce_plan: same counts, same answer)
    splits = batch.w[0].plan->splits[bz];
  }
  const double part = wave_sum_d(nce_finish_row<LPR>(w, a, n, wave * G + g, sub, s_scr[threadIdx.x >> 6], splits));
  // ---- loss: one partial per workgroup; the workgroup that arrives last folds them in order
  __shared__ double wg_part[4];
  if (lane == 0) wg_part[threadIdx.x >> 6] = part;
  __syncthreads();
  const int n_wgs = n_waves / 4;                      // np is a multiple of 64: whole workgroups only
  if (threadIdx.x < 64) {
    if (lane == 0) store_f64_sc1(w.losspart + bx, (wg_part[0] + wg_part[1]) + (wg_part[2] + wg_part[3]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int ticket = 0;
    if (lane == 0) ticket = __hip_atomic_fetch_add(w.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    if (ticket == n_wgs - 1) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      if (lane == 0) __hip_atomic_store(w.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      double t = 0.0;
      for (int k = lane; k < n_wgs; k += 64) t += w.losspart[k];
      t = wave_sum_d(t);
      if (lane == 0) atomicAdd(a.loss, (double)a.loss_scale * t / (double)n);   // one per problem
    }
  }
}

template <int LPR>
__global__ __launch_bounds__(256) void nce_finish_both(NceBatch batch, NceFinishArgs a) {
  nce_finish_both_body<LPR>(batch, a, blockIdx.x, blockIdx.z);
}
template <int LPR>
__global__ __launch_bounds__(256) void nce_finish_bpr2(NceBatch batch, NceFinishArgs a, BprArgs bpr, int n_bpr, int fbx) {
  if ((int)blockIdx.x < n_bpr) { bpr_phase2_body<LPR>(bpr, blockIdx.x); return; }
  const unsigned k = blockIdx.x - n_bpr;
  nce_finish_both_body<LPR>(batch, a, k % fbx, k / fbx);
}

// ---- the fixed-order finish: one row group per touched row, no float atomics ---------------------------------------------
// The reference's backward of `emb[idx]` (XSimGCL.py:30, 35-36) is index_put(accumulate): on one CPU thread the same sum every
// run.  Here the batch's rows are known before the step (srh_sampler_epoch_segments: sorted unique users | positive items |
// other negatives, each with the list of (slot, role) entries that name it), so the LPR lanes of ONE row group own a row:
// they finish its InfoNCE gradients (if a problem names the row), walk its slot list in order adding the BPR / L2 terms, and
// write the row once -- read, add, store.  Same arithmetic per term as the atomic form above; the sum's order is the list's.
#ifndef SRH_RF_SKIP          // (laboratory builds, tools/spmm_lab/build_alt.sh: bit 0 no InfoNCE rows, 1 no slot lists, 2 no loss fold, 3 no final
#define SRH_RF_SKIP 0        //  fold only -- what each part of rows_finish costs, tools/rf_ab.sh; the product is built with 0)
#endif
struct SegArgs {
  const int32_t *n_uniq_u, *n_uniq_i, *n_uniq_n, *rows, *seg_end, *seg, *seg_a, *seg_b, *batch_no;
  int nce_rows, rows_are_zero;
};

// row += v (ZERO: the caller guarantees the row holds zeros -- a plain store)
template <bool ZERO>
__device__ __forceinline__ void put_row(float* table, int row, int lpr, int sub, float4 v) {
  float4* p = reinterpret_cast<float4*>(table) + (size_t)row * lpr + sub;
  if constexpr (ZERO) *p = v;
  else *p = f4_add(*p, v);
}

// The launch is a chain of dependent memory round trips for O(batch) bytes, so the lists are laid out to keep the chain short:
//   1. counts, the group's table row and its list bounds            (nothing depends on the counts for an ADDRESS)
//   2. the list's entries with the rows their terms read, the row's regulariser operand, the InfoNCE partials of the row
//   3. the operand rows and the slots' coefficients
//   4. the store
template <int LPR, bool ZERO>
__device__ __forceinline__ void rows_finish_body(const NceBatch& batch, const NceFinishArgs& fa, const BprArgs& a,
                                                 const SegArgs& sg) {
  constexpr int G = 64 / LPR, GW = 4 * G;         // row groups per wave / per workgroup
  const int rows = a.d_n_rows ? min(*a.d_n_rows, a.B) : a.B;
  if (rows <= 0) return;
  const int lane = threadIdx.x & 63, sub = lane % LPR;
  const int gw = (int)(threadIdx.x >> 6) * G + lane / LPR;          // group within the workgroup
  const int grp = (int)blockIdx.x * GW + gw;
  const int bno = sg.batch_no ? *sg.batch_no : 0;
  const size_t off1 = sg.batch_no ? (size_t)bno * a.B : 0, off3 = 3 * off1;
  const int row = grp < 3 * a.B ? sg.rows[off3 + grp] : -1;
  const int e1 = grp < 3 * a.B ? sg.seg_end[off3 + grp] : 0;
  const int e0g = (grp > 0 && grp < 3 * a.B) ? sg.seg_end[off3 + grp - 1] : 0;
  const int nuu = *sg.n_uniq_u, nui = *sg.n_uniq_i;
  const int kind = grp < nuu ? 0 : (grp < nuu + nui ? 1 : 2);       // user | positive item | negative only
  const float* reg_t = kind == 0 ? a.reg_user : a.reg_item;
  const float* opd_t = kind == 0 ? a.item : a.user;                  // the table a term's operand rows come from
  // ---- what hangs off the group's row and list bounds (the loads above): its regulariser operand and the first entries of
  // its slot list.  Requested from INSIDE the InfoNCE row finish, after that row's own loads -- which hang off nothing but
  // the counts -- have gone out: the two chains run side by side instead of one behind the other.
  constexpr int WU = 8, WI = 8;                   // entries in flight per round trip: user groups / item groups
  int ent[WI], ra[WI], rb[WU];
  auto fetch_entries = [&](int e, auto wc, auto userc) {
    constexpr int W = decltype(wc)::value;
#pragma unroll
    for (int k = 0; k < W; ++k) {
      const int ek = min(e + k, e1 - 1);
      ent[k] = sg.seg[off3 + ek];
      ra[k] = sg.seg_a[off3 + ek];
      if constexpr (decltype(userc)::value) rb[k] = sg.seg_b[off1 + ek];
    }
  };
  bool live = false, any = false;
  int e0 = 0;
  float4 rr = f4_zero();
  auto on_row = [&]() {
    live = row >= 0;
    e0 = live ? e0g : e1;
    rr = reinterpret_cast<const float4*>(reg_t)[(size_t)(live ? row : 0) * LPR + sub];
    any = e0 < e1 && !(SRH_RF_SKIP & 2);
    if (any) {
      if (kind == 0) fetch_entries(e0, std::integral_constant<int, WU>{}, std::true_type{});
      else fetch_entries(e0, std::integral_constant<int, WI>{}, std::false_type{});
    }
  };
  // ---- the InfoNCE gradients of this row (the problem that names it: SegArgs::nce_rows)
  int pz = -1, pi = 0;
  if (sg.nce_rows == 1 && kind < 2) { pz = kind; pi = kind == 0 ? grp : grp - nuu; }
  if (sg.nce_rows == 2 && kind < 2) { pz = 0; pi = grp; }
  const bool has_nce = grp < nuu + nui && pz >= 0 && pz < batch.count;      // (the groups below nuu + nui all have a row)
  float4 dv1 = f4_zero(), dv2 = f4_zero();
  double li = 0.0;
  const bool with_nce = batch.count > 0 && sg.nce_rows != 0;
  // (workgroup-uniform: the groups past the positive items -- other negatives, the dead tail of the 3 B slots: 40 % of the
  //  grid at the Yelp2018 shape -- name no InfoNCE row; their workgroups skip the row finish and report no loss partial)
  const int nce_wgs = with_nce ? (nuu + nui + GW - 1) / GW : 0;
  const bool wg_nce = (int)blockIdx.x < nce_wgs;
  if (wg_nce && !(SRH_RF_SKIP & 1)) {
    const int pq = has_nce ? pz : 0;
    const NceWs& w = batch.w[pq];
    const int n = w.d_n ? min(*w.d_n, w.n_max) : w.n_max;
    int splits = batch.splits;
    if (batch.slots > 0) splits = batch.w[0].plan->splits[pq];
    li = nce_finish_row_grads<LPR>(w, fa, has_nce ? n : 0, pi, sub, splits, dv1, dv2, on_row);
  } else {
    on_row();
  }
  // ---- InfoNCE loss, part 1: one partial per workgroup and problem, summed inside the workgroup in group order and PUBLISHED
  // as soon as it exists (the loss terms are done long before the slot lists are): one 8-byte write-through store into a slot
  // the prep kernel left at "unreported".  The launch's last workgroup -- whose groups are the dead tail of the 3 B slots --
  // reads the slots until none is unreported and folds them in workgroup order while the others are still at their lists:
  // no counter, no wait in any other workgroup, the fold off the launch's critical path.  (A ticket drawn by every workgroup
  // from ONE counter, answer awaited: 384 same-address atomics serialise at ~11 ns each, 2.3 us in front of the lists; the
  // last arrival then folding: 2.8 us behind them -- profiles/r06_c_rows_finish_parts.txt.)
  __shared__ double s_li[GW];
  __shared__ int s_pz[GW];
  const bool fold_loss = with_nce && !(SRH_RF_SKIP & 4);
  if (fold_loss && wg_nce && sub == 0) { s_li[gw] = li; s_pz[gw] = has_nce ? pz : -1; }
  // ---- the fold of BPR phase 1's partials (every workgroup: the regulariser's gradient needs the three norms)
  __shared__ double s_tot[4];
  if (threadIdx.x >= 64 && threadIdx.x < 128) {
    const int t = threadIdx.x - 64;
    double t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    for (int k = t; k < a.n_blocks; k += 64) {
      t0 += a.part[(size_t)k * 4 + 0]; t1 += a.part[(size_t)k * 4 + 1];
      t2 += a.part[(size_t)k * 4 + 2]; t3 += a.part[(size_t)k * 4 + 3];
    }
    t0 = wave_sum_d(t0); t1 = wave_sum_d(t1); t2 = wave_sum_d(t2); t3 = wave_sum_d(t3);
    if (t == 0) { s_tot[0] = t0; s_tot[1] = t1; s_tot[2] = t2; s_tot[3] = t3; }
  }
  __syncthreads();
  const bool folder = blockIdx.x == gridDim.x - 1;
  if (fold_loss && wg_nce && threadIdx.x < 64) {          // wave 0 (the BPR fold above ran on wave 1)
    if ((int)threadIdx.x < batch.count) {                 // problem k's partial of this workgroup: its losspart[blockIdx.x]
      double t = 0.0;
      for (int k = 0; k < GW; ++k)
        if (s_pz[k] == (int)threadIdx.x) t += s_li[k];
      store_f64_sc1(batch.w[threadIdx.x].losspart + blockIdx.x, t);
    }
  }
  const float nu = (float)sqrt(s_tot[1]), np = (float)sqrt(s_tot[2]), nn = (float)sqrt(s_tot[3]);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float r = nu / (float)rows + np / (float)rows;
    if (a.reg_include_neg) r += nn / (float)rows;
    a.losses[0] += (double)a.loss_scale * s_tot[0] / (double)rows;
    a.losses[1] += (double)(a.loss_scale * (r * a.reg_coef));
  }
  const float cs = a.loss_scale / (float)rows;
  const float rs = a.reg_coef * a.loss_scale / (float)rows;
  const float cu = nu > 0.f ? rs / nu : 0.f, cp = np > 0.f ? rs / np : 0.f;
  const float cn = (a.reg_include_neg && nn > 0.f) ? rs / nn : 0.f;
  const bool same_u = (a.reg_user == a.user) && (a.greg_user == a.g_user);
  const bool same_i = (a.reg_item == a.item) && (a.greg_item == a.g_item);
  const bool same = kind == 0 ? same_u : same_i;
  // ---- the row's slot list, summed in list order.  One round trip per W entries: a popular item is named by ~30 slots of a
  // batch of 2048 (its group is the launch's critical path), so a group keeps the operand rows of 8 entries in flight
  // (more costs the launch its second wave per SIMD: 256 VGPRs); 92 % of the groups have one entry.
  float4 acc = f4_zero(), accr = f4_zero();
  auto sum_list = [&](auto wc, auto userc) {
    constexpr int W = decltype(wc)::value;
    constexpr bool USER = decltype(userc)::value;
    for (int e = e0; e < e1; e += W) {
      float4 x[W], y[USER ? W : 1];
      float cf[W];
      int role[W];
#pragma unroll
      for (int k = 0; k < W; ++k) {
        x[k] = reinterpret_cast<const float4*>(opd_t)[(size_t)ra[k] * LPR + sub];
        if constexpr (USER) y[k] = reinterpret_cast<const float4*>(opd_t)[(size_t)rb[k] * LPR + sub];
        cf[k] = a.coef[ent[k] >> 2];
        role[k] = ent[k] & 3;
      }
      if (e + W < e1) fetch_entries(e + W, wc, userc);
#pragma unroll
      for (int k = 0; k < W; ++k) {
        if (e + k >= e1) break;
        const float c = cf[k] * cs;
        float4 g;
        float cr;
        if constexpr (USER) {
          g = make_float4(c * (x[k].x - y[k].x), c * (x[k].y - y[k].y), c * (x[k].z - y[k].z), c * (x[k].w - y[k].w));
          cr = cu;
        } else if (role[k] == 1) {
          g = f4_scale(x[k], c);
          cr = cp;
        } else {
          g = f4_scale(x[k], -c);
          cr = cn;
        }
        if (same) g = f4_fma(cr, rr, g);
        else accr = f4_fma(cr, rr, accr);
        acc = f4_add(acc, g);
      }
    }
  };
  if (any) {
    if (kind == 0) sum_list(std::integral_constant<int, WU>{}, std::true_type{});
    else sum_list(std::integral_constant<int, WI>{}, std::false_type{});
  }
  // ---- one write per table the row belongs to
  if (live) {
    float* tb = kind == 0 ? a.g_user : a.g_item;
    if (has_nce) {
      const NceWs& w = batch.w[pz];
      if (w.g1 == tb) acc = f4_add(acc, dv1);
      if (w.g2 == tb) acc = f4_add(acc, dv2);
      if (w.g1 != tb) put_row<ZERO>(w.g1, row, LPR, sub, w.g2 == w.g1 ? f4_add(dv1, dv2) : dv1);
      if (w.g2 != tb && w.g2 != w.g1) put_row<ZERO>(w.g2, row, LPR, sub, dv2);
    }
    put_row<ZERO>(tb, row, LPR, sub, acc);
    if (!same) put_row<ZERO>(kind == 0 ? a.greg_user : a.greg_item, row, LPR, sub, accr);
  }
  // ---- InfoNCE loss, part 2: the last workgroup of the grid reads the slots until every workgroup has reported (they were
  // all dispatched before it: nothing it waits for can be waiting for it) and folds them in workgroup order, a wave per problem
  if (!fold_loss || (SRH_RF_SKIP & 8) || !folder) return;
  const int k = (int)(threadIdx.x >> 6);
  if (k < batch.count) {
    const NceWs& w = batch.w[k];
    const int n = w.d_n ? min(*w.d_n, w.n_max) : w.n_max;
    const unsigned long long* slot = reinterpret_cast<const unsigned long long*>(w.losspart);
    double t = 0.0;
    for (int j0 = 0; j0 < nce_wgs; j0 += 512) {           // (eight slots per lane in flight: one round trip per poll)
      unsigned long long v[8];
      bool missing = false;
      for (int spin = 0; spin < (1 << 16); ++spin) {      // (bounded: a lost report never hangs the device ...)
        missing = false;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int j = j0 + 64 * q + lane;
          v[q] = j < nce_wgs ? __hip_atomic_load(slot + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
          missing = missing || v[q] == kLossUnreported;
        }
        if (__builtin_amdgcn_ballot_w64(missing) == 0) break;
        __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) t += __builtin_bit_cast(double, v[q]);    // (... and shows: an unreported slot is a NaN pattern)
    }
    t = wave_sum_d(t);
    s_li[k] = n > 0 ? (double)fa.loss_scale * t / (double)n : 0.0;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int j = 0; j < batch.count; ++j) t += s_li[j];
    atomicAdd(fa.loss, t);
  }
}

template <int LPR, bool ZERO>
__global__ __launch_bounds__(256) void rows_finish(NceBatch batch, NceFinishArgs fa, BprArgs bpr, SegArgs sg) {
  rows_finish_body<LPR, ZERO>(batch, fa, bpr, sg);
}
template <int LPR>
void launch_rows_finish(const NceBatch& batch, const NceFinishArgs& fa, const BprArgs& bp, const SegArgs& sg, hipStream_t st) {
  constexpr int G = 64 / LPR;
  const int grid = (3 * bp.B + 4 * G - 1) / (4 * G);
  if (sg.rows_are_zero) rows_finish<LPR, true><<<grid, 256, 0, st>>>(batch, fa, bp, sg);
  else rows_finish<LPR, false><<<grid, 256, 0, st>>>(batch, fa, bp, sg);
}

template <int D>
srh_status_t launch_infonce(const srh_infonce_problem_t* pr, int count, float tau, float loss_scale, double* loss,
                            void* ws, hipStream_t st, int precision, const BprArgs* bpr = nullptr,
                            const SegArgs* seg = nullptr) {
  constexpr int LPR = D / 4, G = 64 / LPR;
  NceBatch batch{};
  batch.count = count;
  batch.splits = kNceUsedSplits;
  int np_max = 0;
  char* cursor = reinterpret_cast<char*>(ws);
  for (int k = 0; k < count; ++k) {
    NceWs w = carve_nce(cursor, pr[k].n, D);
    cursor += srh_infonce_ws_bytes(pr[k].n, D);
    w.src1 = pr[k].d_v1; w.src2 = pr[k].d_v2; w.idx = pr[k].d_idx; w.d_n = pr[k].d_n; w.n_max = (int)pr[k].n;
    w.g1 = pr[k].d_g1; w.g2 = pr[k].d_g2;
    w.g2_plain = pr[k].g2_exclusive != 0 && pr[k].d_g2 != pr[k].d_g1;
    batch.w[k] = w;
    np_max = std::max(np_max, (int)w.np);
  }
  const float inv_tau = 1.0f / tau;
  const bool f32 = precision == SRH_NCE_F32;
  int grid_f32 = 0;
  if (f32) {
    if constexpr (D > 128) {
      srh::set_error("infonce_fwd_bwd: the all-f32 MFMA path serves d = 64 / 128");
      return SRH_ERR_UNSUPPORTED;
    }
    // one persistent workgroup per CU (the 80+ KB ring admits one); never more than the largest cut nce_plan can make
    int64_t most = 0;
    for (int k = 0; k < count; ++k) {
      if (batch.w[k].np / kNceKeyBlock > (int64_t)kNceSplits * kNceMaxPer) {
        srh::set_error("infonce_fwd_bwd: the all-f32 MFMA path serves n <= %d", kNceSplits * kNceMaxPer * kNceKeyBlock);
        return SRH_ERR_UNSUPPORTED;
      }
      most += (batch.w[k].np + kNceQueryBlock - 1) / kNceQueryBlock * kNceSplits;
    }
    batch.slots = srh::cu_count();
    batch.f32_only = 1;
    grid_f32 = (int)std::min<int64_t>(batch.slots, most);
  }
  dim3 gp((np_max / G + 3) / 4, 2, count);
  BprArgs bp{};
  int n_bpr = 0;
  if (bpr) {
    bp = *bpr;
    n_bpr = bp.n_blocks = ((bp.B + G - 1) / G + 3) / 4;
    nce_prep_bpr1<LPR><<<n_bpr + (int)(gp.x * 2 * count), 256, 0, st>>>(batch, bp, n_bpr, (int)gp.x);
  } else {
    nce_prep<LPR><<<gp, 256, 0, st>>>(batch);
  }
  SRH_LAUNCH_CHECK();
  NceFinishArgs fa{inv_tau, loss_scale, loss};
  dim3 fb((np_max / G + 3) / 4, 1, count);
  if (!f32) {
    constexpr int kLds = nce_lds_bytes<D, kNcePvTerms>();
    static const bool attr_set = [] {
      (void)hipFuncSetAttribute((const void*)nce_tile_lds<D, false, 1, 8, kNcePvTerms>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
      (void)hipFuncSetAttribute((const void*)nce_tile_lds<D, true, 1, 8, kNcePvTerms>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
      return true;
    }();
    (void)attr_set;
    dim3 gl((np_max + 127) / 128, batch.splits, count);          // 8 waves x 16 queries per workgroup
    nce_tile_lds<D, false, 1, 8, kNcePvTerms><<<gl, 512, kLds, st>>>(batch, inv_tau);
    SRH_LAUNCH_CHECK();
    nce_tile_lds<D, true, 1, 8, kNcePvTerms><<<gl, 512, kLds, st>>>(batch, inv_tau);
    SRH_LAUNCH_CHECK();
  } else if constexpr (D <= 128) {
    // ---- SRH_NCE_F32: both products on v_mfma_f32_16x16x4_f32 (exact f32 multiply-adds), same four launches ----
    constexpr int kLds = NceF32<D>::kLds;
    static const bool attr_set = [] {
      (void)hipFuncSetAttribute((const void*)nce_tile_f32<D, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
      (void)hipFuncSetAttribute((const void*)nce_tile_f32<D, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
      (void)hipFuncSetAttribute((const void*)nce_tile_f32<D, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
      (void)hipFuncSetAttribute((const void*)nce_tile_f32<D, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
      return true;
    }();
    (void)attr_set;
    bool reuse = true;                     // pass 1 keeps its weights for pass 2 (every problem has the n x n array)
    for (int k = 0; k < count; ++k) reuse = reuse && batch.w[k].etile != nullptr;
    if (reuse) {
      nce_tile_f32<D, false, true><<<grid_f32, 512, kLds, st>>>(batch, inv_tau);
      SRH_LAUNCH_CHECK();
      nce_tile_f32<D, true, true><<<grid_f32, 512, kLds, st>>>(batch, inv_tau);
    } else {
      nce_tile_f32<D, false, false><<<grid_f32, 512, kLds, st>>>(batch, inv_tau);
      SRH_LAUNCH_CHECK();
      nce_tile_f32<D, true, false><<<grid_f32, 512, kLds, st>>>(batch, inv_tau);
    }
    SRH_LAUNCH_CHECK();
  }
  if (bpr && seg) {
    for (int k = 0; k < count; ++k)             // (one loss partial per workgroup of the finish in every problem's losspart array)
      SRH_REQUIRE(batch.w[k].np >= (3 * bp.B + 4 * G - 1) / (4 * G),
                  "bpr_infonce_fwd_bwd: with batch segments every InfoNCE problem must be sized for the batch (n >= ~3 B d / 256)");
    launch_rows_finish<LPR>(batch, fa, bp, *seg, st);
  }
  else if (bpr) nce_finish_bpr2<LPR><<<n_bpr + (int)(fb.x * count), 256, 0, st>>>(batch, fa, bp, n_bpr, (int)fb.x);
  else nce_finish_both<LPR><<<fb, 256, 0, st>>>(batch, fa);
  SRH_LAUNCH_CHECK();
  return SRH_OK;
}

template <int LPR>
srh_status_t launch_bpr(const BprArgs& a, hipStream_t st, const SegArgs* seg = nullptr) {
  constexpr int G = 64 / LPR;
  const int blocks = ((a.B + G - 1) / G + 3) / 4;
  BprArgs b = a;
  b.n_blocks = blocks;
  bpr_phase1<LPR><<<blocks, 256, 0, st>>>(b);
  SRH_LAUNCH_CHECK();
  if (seg) {
    NceBatch none{};
    launch_rows_finish<LPR>(none, NceFinishArgs{}, b, *seg, st);
  } else {
    bpr_phase2<LPR><<<blocks, 256, 0, st>>>(b);
  }
  SRH_LAUNCH_CHECK();
  return SRH_OK;
}

// srh_batch_segments_t -> SegArgs (checked)
static srh_status_t seg_args(const srh_batch_segments_t* g, int n_problems, SegArgs& out) {
  SRH_REQUIRE(g->d_n_uniq_u && g->d_n_uniq_i && g->d_n_uniq_n && g->d_seg_rows && g->d_seg_end && g->d_seg && g->d_seg_a && g->d_seg_b,
              "batch segments: null array");
  SRH_REQUIRE(g->nce_rows >= 0 && g->nce_rows <= 2, "batch segments: nce_rows must be 0, 1 or 2");
  SRH_REQUIRE(g->nce_rows != 1 || n_problems == 2, "batch segments: nce_rows = 1 names two InfoNCE problems (users, positive items)");
  SRH_REQUIRE(g->nce_rows != 2 || n_problems == 1, "batch segments: nce_rows = 2 names one InfoNCE problem ([users ; positive items])");
  SRH_REQUIRE(g->nce_rows != 0 || n_problems == 0,
              "batch segments: InfoNCE problems in the call need nce_rows 1 or 2 (their rows are finished by the row groups)");
  out = SegArgs{g->d_n_uniq_u, g->d_n_uniq_i, g->d_n_uniq_n, g->d_seg_rows, g->d_seg_end, g->d_seg, g->d_seg_a, g->d_seg_b,
                g->d_batch_no, g->nce_rows, g->rows_are_zero};
  return SRH_OK;
}

}  // namespace

extern "C" {

// 32 B per phase-1 workgroup (at most B/4 + 1 of them) followed by one float per batch row
static inline int64_t bpr_part_bytes(int64_t B) { return 32 * ((B > 0 ? B : 0) / 4 + 2); }
int64_t srh_bpr_ws_bytes(int64_t B) { return bpr_part_bytes(B) + 4 * (B > 0 ? B : 0) + 64; }

srh_status_t srh_bpr_l2_fwd_bwd(const float* d_user, const float* d_item, const float* d_reg_user,
                                const float* d_reg_item, const int32_t* d_u_idx, const int32_t* d_i_idx,
                                const int32_t* d_j_idx, int64_t B, const int32_t* d_n_rows, int32_t d,
                                float reg_coef, int32_t reg_include_neg, float loss_scale, float* d_g_user,
                                float* d_g_item, float* d_greg_user, float* d_greg_item, double* d_losses,
                                void* d_ws, void* stream) {
  SRH_REQUIRE(d_user && d_item && d_reg_user && d_reg_item && d_u_idx && d_i_idx && d_j_idx, "bpr_l2_fwd_bwd: null input");
  SRH_REQUIRE(d_g_user && d_g_item && d_greg_user && d_greg_item && d_losses && d_ws, "bpr_l2_fwd_bwd: null output");
  SRH_REQUIRE(B > 0 && B < (int64_t(1) << 30), "bpr_l2_fwd_bwd: bad batch size");
  SRH_REQUIRE(srh::dim_supported(d), "bpr_l2_fwd_bwd: d=%d unsupported", d);
  BprArgs a{d_user, d_item, d_reg_user, d_reg_item, d_u_idx, d_i_idx, d_j_idx, d_n_rows, (int)B,
            reg_coef, loss_scale, reg_include_neg, d_g_user, d_g_item, d_greg_user, d_greg_item, d_losses,
            reinterpret_cast<double*>(d_ws), reinterpret_cast<float*>(reinterpret_cast<char*>(d_ws) + bpr_part_bytes(B)), 0};
  hipStream_t st = srh::as_stream(stream);
  switch (d) {
    case 32: return launch_bpr<8>(a, st);
    case 64: return launch_bpr<16>(a, st);
    case 128: return launch_bpr<32>(a, st);
    default: return launch_bpr<64>(a, st);
  }
}

srh_status_t srh_bpr_fwd(const float* d_u, const float* d_p, const float* d_n, int64_t B, int32_t d,
                         void* d_scalar_ws, float* d_loss, float* d_coef, void* stream) {
  SRH_REQUIRE(d_u && d_p && d_n && d_scalar_ws && d_loss && d_coef, "bpr_fwd: null argument");
  SRH_REQUIRE(B > 0 && B < (int64_t(1) << 30), "bpr_fwd: bad batch size");
  SRH_REQUIRE(srh::dim_supported(d), "bpr_fwd: d=%d unsupported", d);
  hipStream_t st = srh::as_stream(stream);
  const float4 *U = reinterpret_cast<const float4*>(d_u), *P = reinterpret_cast<const float4*>(d_p),
               *N = reinterpret_cast<const float4*>(d_n);
  ScalarWs* ws = reinterpret_cast<ScalarWs*>(d_scalar_ws);
#define SRH_BPR_FWD(LPR)                                                                       \
  bpr_plain_fwd<LPR><<<(int)(((B + 64 / LPR - 1) / (64 / LPR) + 3) / 4), 256, 0, st>>>(U, P, N, (int)B, ws, d_loss, d_coef)
  switch (d) {
    case 32: SRH_BPR_FWD(8); break;
    case 64: SRH_BPR_FWD(16); break;
    case 128: SRH_BPR_FWD(32); break;
    default: SRH_BPR_FWD(64); break;
  }
#undef SRH_BPR_FWD
  SRH_LAUNCH_CHECK();
  return SRH_OK;
}

srh_status_t srh_bpr_bwd(const float* d_u, const float* d_p, const float* d_n, const float* d_coef,
                         int64_t B, int32_t d, const float* d_gout, float* d_gu, float* d_gp, float* d_gn, void* stream) {
  SRH_REQUIRE(d_u && d_p && d_n && d_coef && d_gout && d_gu && d_gp && d_gn, "bpr_bwd: null argument");
  SRH_REQUIRE(B > 0 && B < (int64_t(1) << 30), "bpr_bwd: bad batch size");
  SRH_REQUIRE(srh::dim_supported(d), "bpr_bwd: d=%d unsupported", d);
  hipStream_t st = srh::as_stream(stream);
  const int64_t threads = B * (d / 4);
  const int blocks = (int)((threads + 255) / 256);
  const float4 *U = reinterpret_cast<const float4*>(d_u), *P = reinterpret_cast<const float4*>(d_p),
               *N = reinterpret_cast<const float4*>(d_n);
  float4 *GU = reinterpret_cast<float4*>(d_gu), *GP = reinterpret_cast<float4*>(d_gp), *GN = reinterpret_cast<float4*>(d_gn);
  switch (d) {
    case 32: bpr_plain_bwd<8><<<blocks, 256, 0, st>>>(U, P, N, d_coef, (int)B, d_gout, GU, GP, GN); break;
    case 64: bpr_plain_bwd<16><<<blocks, 256, 0, st>>>(U, P, N, d_coef, (int)B, d_gout, GU, GP, GN); break;
    case 128: bpr_plain_bwd<32><<<blocks, 256, 0, st>>>(U, P, N, d_coef, (int)B, d_gout, GU, GP, GN); break;
    default: bpr_plain_bwd<64><<<blocks, 256, 0, st>>>(U, P, N, d_coef, (int)B, d_gout, GU, GP, GN); break;
  }
  SRH_LAUNCH_CHECK();
  return SRH_OK;
}

static srh_status_t l2_args(const srh_l2_block_t* blocks, int32_t n_blocks, float reg, bool backward, L2Args& a) {
  SRH_REQUIRE(blocks && n_blocks >= 1 && n_blocks <= kL2Blocks, "l2_reg: 1..%d blocks of rows per call", kL2Blocks);
  a.count = n_blocks;
  a.reg = reg;
  a.first_wg[0] = 0;
  for (int k = 0; k < n_blocks; ++k) {
    const srh_l2_block_t& b = blocks[k];
    SRH_REQUIRE(b.d_x && b.rows > 0 && b.cols > 0, "l2_reg: block %d is empty or null", k);
    SRH_REQUIRE(!backward || b.d_gx, "l2_reg_bwd: block %d has no gradient buffer", k);
    a.x[k] = b.d_x;
    a.gx[k] = b.d_gx;
    a.n[k] = b.rows * b.cols;
    a.rows[k] = (float)b.rows;
    a.first_wg[k + 1] = a.first_wg[k] + (int)std::min<int64_t>((a.n[k] + 1023) / 1024, kL2MaxWg);
  }
  return SRH_OK;
}

srh_status_t srh_l2_reg_fwd(const srh_l2_block_t* blocks, int32_t n_blocks, float reg, void* d_scalar_ws,
                            float* d_norms, float* d_loss, void* stream) {
  SRH_REQUIRE(d_scalar_ws && d_norms && d_loss, "l2_reg_fwd: null argument");
  L2Args a{};
  if (srh_status_t s = l2_args(blocks, n_blocks, reg, false, a)) return s;
  l2_reg_fwd_kernel<<<a.first_wg[n_blocks], 256, 0, srh::as_stream(stream)>>>(a, reinterpret_cast<ScalarWs*>(d_scalar_ws),
                                                                             d_norms, d_loss);
  SRH_LAUNCH_CHECK();
  return SRH_OK;
}

srh_status_t srh_l2_reg_bwd(const srh_l2_block_t* blocks, int32_t n_blocks, float reg, const float* d_norms,
                            const float* d_gout, void* stream) {
  SRH_REQUIRE(d_norms && d_gout, "l2_reg_bwd: null argument");
  L2Args a{};
  if (srh_status_t s = l2_args(blocks, n_blocks, reg, true, a)) return s;
  l2_reg_bwd_kernel<<<a.first_wg[n_blocks], 256, 0, srh::as_stream(stream)>>>(a, d_norms, d_gout);
  SRH_LAUNCH_CHECK();
  return SRH_OK;
}

int64_t srh_infonce_ws_bytes(int64_t n, int32_t d) {
  if (n <= 0 || d <= 0) return 0;
  const int64_t np = nce_pad(n);
  return 4 * (2 * np * d + 2 * (int64_t)kNceSplits * np * d + 4 * np + (int64_t)kNceSplits * np) + 8 * np +
         20 * np * d + 4 * (np / 16) + 512 +      // (2 views x 5 sixteen-bit operand images; counters; the plan record)
         (np <= kNceReuseMax ? 4 * np * np : 0);  // (pass 1's weights for pass 2: NceWs::etile)
}

static srh_status_t infonce_entry(const srh_infonce_problem_t* problems, int32_t n_problems, int32_t d, float tau,
                                  float loss_scale, double* d_loss, void* d_ws, int32_t precision, void* stream,
                                  const BprArgs* bpr, const SegArgs* seg = nullptr) {
  SRH_REQUIRE(precision == SRH_NCE_DEFAULT || precision == SRH_NCE_SPLIT16 || precision == SRH_NCE_F32,
              "infonce_fwd_bwd: unknown precision %d", precision);
  if (precision == SRH_NCE_DEFAULT) {
    precision = g_nce_precision.load(std::memory_order_relaxed);
    if (d == 256 && precision == SRH_NCE_F32) {
      // the f32 passes serve d = 64 / 128.  The default must not fail on d, but it must not change arithmetic silently
      // either: said once per process, on stderr (an explicit SRH_NCE_F32 at d = 256 is refused below)
      static std::atomic<bool> said{false};
      if (!said.exchange(true))
        fprintf(stderr, "[selfrec_hip] infonce: d = 256 has no all-f32 MFMA path; SRH_NCE_DEFAULT resolves to SRH_NCE_SPLIT16 "
                        "(f16 hi+lo / bf16 hi+mid operands, f32 accumulation: logits to 2^-22) for these calls\n");
      precision = SRH_NCE_SPLIT16;
    }
  }
  SRH_REQUIRE(problems && d_loss && d_ws, "infonce_fwd_bwd: null argument");
  SRH_REQUIRE(n_problems >= 1 && n_problems <= kNceMaxProblems, "infonce_fwd_bwd: 1..%d problems per call", kNceMaxProblems);
  SRH_REQUIRE(d == 64 || d == 128 || d == 256, "infonce_fwd_bwd: d=%d unsupported (need 64, 128 or 256)", d);
  SRH_REQUIRE(d != 256 || precision != SRH_NCE_F32,
              "infonce_fwd_bwd: the all-f32 MFMA path serves d = 64 / 128 (d = 256: the split path only)");
  for (int k = 0; k < n_problems; ++k) {
    const srh_infonce_problem_t& p = problems[k];
    SRH_REQUIRE(p.d_v1 && p.d_v2 && p.d_g1 && p.d_g2, "infonce_fwd_bwd: null tensor in problem %d", k);
    SRH_REQUIRE(p.n > 0 && p.n < (int64_t(1) << 24), "infonce_fwd_bwd: bad n in problem %d", k);
  }
  if (!(tau >= 0.03f)) {
    srh::set_error("infonce_fwd_bwd: temperature %g below 0.03 needs a running max (not implemented)", (double)tau);
    return SRH_ERR_UNSUPPORTED;
  }
  hipStream_t st = srh::as_stream(stream);
  if (d == 64) return launch_infonce<64>(problems, n_problems, tau, loss_scale, d_loss, d_ws, st, precision, bpr, seg);
  if (d == 128) return launch_infonce<128>(problems, n_problems, tau, loss_scale, d_loss, d_ws, st, precision, bpr, seg);
  return launch_infonce<256>(problems, n_problems, tau, loss_scale, d_loss, d_ws, st, precision, bpr, seg);
}

#ifdef SRH_NCEF32_STAMPS
srh_status_t srh_debug_nce_stamps(unsigned long long* h_out /* 2 x 512 x 8 x 8 */) {
  SRH_HIP(hipDeviceSynchronize());
  SRH_HIP(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_nce_stamps), sizeof(g_nce_stamps)));
  return SRH_OK;
}
#endif

srh_status_t srh_infonce_set_precision(int32_t mode) {
  SRH_REQUIRE(mode == SRH_NCE_SPLIT_BF16 || mode == SRH_NCE_F32, "infonce_set_precision: unknown mode %d", mode);
  g_nce_precision.store(mode, std::memory_order_relaxed);
  return SRH_OK;
}
int32_t srh_infonce_get_precision(void) { return g_nce_precision.load(std::memory_order_relaxed); }

srh_status_t srh_infonce_fwd_bwd_multi(const srh_infonce_problem_t* problems, int32_t n_problems, int32_t d,
                                       float tau, float loss_scale, double* d_loss, void* d_ws, int32_t precision,
                                       void* stream) {
  return infonce_entry(problems, n_problems, d, tau, loss_scale, d_loss, d_ws, precision, stream, nullptr);
}

srh_status_t srh_bpr_infonce_fwd_bwd(const srh_bpr_problem_t* b, const srh_infonce_problem_t* problems,
                                     int32_t n_problems, int32_t d, float tau, float cl_scale, double* d_cl_loss,
                                     void* d_nce_ws, int32_t precision, void* stream) {
  SRH_REQUIRE(b, "bpr_infonce_fwd_bwd: null bpr problem");
  SRH_REQUIRE(b->d_user && b->d_item && b->d_reg_user && b->d_reg_item && b->d_u_idx && b->d_i_idx && b->d_j_idx,
              "bpr_infonce_fwd_bwd: null input");
  SRH_REQUIRE(b->d_g_user && b->d_g_item && b->d_greg_user && b->d_greg_item && b->d_losses && b->d_ws,
              "bpr_infonce_fwd_bwd: null output");
  SRH_REQUIRE(b->B > 0 && b->B < (int64_t(1) << 30), "bpr_infonce_fwd_bwd: bad batch size");
  BprArgs a{b->d_user, b->d_item, b->d_reg_user, b->d_reg_item, b->d_u_idx, b->d_i_idx, b->d_j_idx, b->d_n_rows,
            (int)b->B, b->reg_coef, b->loss_scale, b->reg_include_neg, b->d_g_user, b->d_g_item, b->d_greg_user,
            b->d_greg_item, b->d_losses, reinterpret_cast<double*>(b->d_ws),
            reinterpret_cast<float*>(reinterpret_cast<char*>(b->d_ws) + bpr_part_bytes(b->B)), 0};
  SegArgs sg{};
  if (b->seg)
    if (srh_status_t st = seg_args(b->seg, n_problems, sg)) return st;
  return infonce_entry(problems, n_problems, d, tau, cl_scale, d_cl_loss, d_nce_ws, precision, stream, &a, b->seg ? &sg : nullptr);
}

srh_status_t srh_bpr_l2_fwd_bwd_p(const srh_bpr_problem_t* b, int32_t d, void* stream) {
  SRH_REQUIRE(b, "bpr_l2_fwd_bwd_p: null bpr problem");
  SRH_REQUIRE(b->d_user && b->d_item && b->d_reg_user && b->d_reg_item && b->d_u_idx && b->d_i_idx && b->d_j_idx,
              "bpr_l2_fwd_bwd_p: null input");
  SRH_REQUIRE(b->d_g_user && b->d_g_item && b->d_greg_user && b->d_greg_item && b->d_losses && b->d_ws,
              "bpr_l2_fwd_bwd_p: null output");
  SRH_REQUIRE(b->B > 0 && b->B < (int64_t(1) << 28), "bpr_l2_fwd_bwd_p: bad batch size");
  SRH_REQUIRE(srh::dim_supported(d), "bpr_l2_fwd_bwd_p: d=%d unsupported", d);
  BprArgs a{b->d_user, b->d_item, b->d_reg_user, b->d_reg_item, b->d_u_idx, b->d_i_idx, b->d_j_idx, b->d_n_rows,
            (int)b->B, b->reg_coef, b->loss_scale, b->reg_include_neg, b->d_g_user, b->d_g_item, b->d_greg_user,
            b->d_greg_item, b->d_losses, reinterpret_cast<double*>(b->d_ws),
            reinterpret_cast<float*>(reinterpret_cast<char*>(b->d_ws) + bpr_part_bytes(b->B)), 0};
  SegArgs sg{};
  if (b->seg)
    if (srh_status_t st = seg_args(b->seg, 0, sg)) return st;
  hipStream_t st = srh::as_stream(stream);
  const SegArgs* sp = b->seg ? &sg : nullptr;
  switch (d) {
    case 32: return launch_bpr<8>(a, st, sp);
    case 64: return launch_bpr<16>(a, st, sp);
    case 128: return launch_bpr<32>(a, st, sp);
    default: return launch_bpr<64>(a, st, sp);
  }
}

srh_status_t srh_infonce_fwd_bwd(const float* d_v1, const float* d_v2, const int32_t* d_idx, int64_t n,
                                 const int32_t* d_n, int32_t d, float tau, float loss_scale, double* d_loss,
                                 float* d_g1, float* d_g2, void* d_ws, void* stream) {
  srh_infonce_problem_t p{d_v1, d_v2, d_idx, n, d_n, d_g1, d_g2};
  return srh_infonce_fwd_bwd_multi(&p, 1, d, tau, loss_scale, d_loss, d_ws, SRH_NCE_DEFAULT, stream);
}

}  // extern "C"
