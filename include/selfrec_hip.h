/*
 * selfrec_hip.h -- C ABI of libselfrec_hip.so, the MI355X (gfx950) engine behind SELFRec's
 * graph-recommender hot path.
 *
 * The reference (Coder-Yu/SELFRec) is pure Python: it has no FFI of its own.  The
 * "operator interface" it does have is a set of Python call sites inside its model
 * files; each entry point below is what a ctypes stub bound at that call site calls
 * (INTEGRATION.md shows the stubs).  Every declaration cites the reference interface it
 * replaces as  <file>:<lines>  relative to the reference checkout.
 *
 * Conventions
 *   - plain C types only; no torch / C++ types cross this boundary.
 *   - pointers named d_*  are raw DEVICE addresses on the current HIP device,
 *     pointers named h_*  are HOST addresses.  All buffers are caller-owned; the library
 *     allocates only the opaque handles it returns (freed by the matching *_destroy).
 *   - every device entry point is asynchronous on the `stream` argument (a hipStream_t
 *     passed as void*); none synchronises, none uses the null stream implicitly.
 *   - return value: SRH_OK (0) or a negative srh_status_t.  Nothing throws, nothing calls
 *     exit().  srh_last_error_string() describes the last failure on the calling thread.
 *   - row-major fp32 matrices with leading dimension == d unless stated.
 *   - handles are not thread-safe; distinct handles may be used from distinct threads.
 */
#ifndef SELFREC_HIP_H
#define SELFREC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t srh_status_t;
enum {
  SRH_OK = 0,
  SRH_ERR_INVALID_ARG = -1,
  SRH_ERR_HIP = -2,
  SRH_ERR_UNSUPPORTED = -3,
  SRH_ERR_NOMEM = -4,
  SRH_ERR_STATE = -5
};

/* ABI version: bumped whenever a signature or struct below changes. */
#define SRH_ABI_VERSION 30
int32_t srh_abi_version(void);
const char* srh_last_error_string(void);
/* Number of visible HIP devices (0 when there is none -- never an error). */
int32_t srh_device_count(void);

/* ------------------------------------------------------------------------------------
 * (a-1) Pairwise sampler -- replaces util/sampler.py:5-28  next_batch_pairwise(data, B, n_negs)
 *
 * Host-side, bit-exact replay of the CPython `random` stream the reference consumes:
 * MT19937 words -> getrandbits(k) -> _randbelow(n) -> shuffle / choice with set rejection.
 * The handle keeps the persistent order of data.training_data across epochs (the
 * reference shuffles that shared list in place, sampler.py:7).
 * ---------------------------------------------------------------------------------- */
typedef struct srh_sampler srh_sampler_t;

/* h_edge_user/h_edge_item: the id columns of training_data in its CURRENT order
 * (ids as assigned by data/ui_graph.py:29-38).  Copied. */
srh_status_t srh_sampler_create(srh_sampler_t** out, int64_t n_users, int64_t n_items,
                                int64_t n_edges, const int32_t* h_edge_user,
                                const int32_t* h_edge_item);
void srh_sampler_destroy(srh_sampler_t* s);
/* MT19937 state exactly as random.getstate()[1]: 624 words followed by the position. */
srh_status_t srh_sampler_set_state(srh_sampler_t* s, const uint32_t* h_mt624, int32_t pos);
srh_status_t srh_sampler_get_state(const srh_sampler_t* s, uint32_t* h_mt624, int32_t* pos);
/* random.seed(int) for 0 <= seed < 2^64 (CPython init_by_array on the 32-bit limbs). */
srh_status_t srh_sampler_seed(srh_sampler_t* s, uint64_t seed);
/* random.shuffle(training_data): one Fisher-Yates pass over the persistent order. */
srh_status_t srh_sampler_shuffle(srh_sampler_t* s);
/* Current order: h_perm[p] = index into the edge arrays given at create time. */
srh_status_t srh_sampler_get_order(const srh_sampler_t* s, int64_t* h_perm);
/* One batch: positions [ptr, ptr+count) of the current order; writes count users, count
 * positives and count*n_negs negatives.  *out_count receives count
 * (= min(batch_size, n_edges-ptr), sampler.py:10-14).  */
srh_status_t srh_sampler_next_batch(srh_sampler_t* s, int64_t ptr, int64_t batch_size,
                                    int32_t n_negs, int32_t* h_u, int32_t* h_i, int32_t* h_j,
                                    int64_t* out_count);
/* A whole epoch = shuffle + every batch, contiguous: h_u/h_i hold n_edges ids, h_j holds
 * n_edges*n_negs.  When h_uniq_u/h_uniq_i are non-NULL they receive, per batch b, the
 * sorted unique user / positive-item ids (torch.unique at XSimGCL.py:46-47) at offset
 * b*batch_size, with the counts in h_n_uniq_u[b] / h_n_uniq_i[b]. */
srh_status_t srh_sampler_epoch(srh_sampler_t* s, int64_t batch_size, int32_t n_negs,
                               int32_t* h_u, int32_t* h_i, int32_t* h_j,
                               int32_t* h_uniq_u, int32_t* h_n_uniq_u,
                               int32_t* h_uniq_i, int32_t* h_n_uniq_i);
/* The row -> slot lists of every batch of an epoch: what lets the batch gradients of XSimGCL.py:30-37 (emb[idx] gathers whose
 * autograd is a scatter-ADD: a user / item that occurs in several pairs of a batch receives several contributions) be summed in
 * ONE FIXED ORDER by one row group each, without float atomics -- a training step is then reproducible bit for bit, like the
 * reference's single-threaded CPU step.  Host only, no draw from the generator; inputs are srh_sampler_epoch's outputs (ids).
 * Per batch b (cnt pairs), the ROW GROUPS are numbered
 *     [0, nuu)                 the sorted unique users                    (h_uniq_u)
 *     [nuu, nuu + nui)         the sorted unique positive items           (h_uniq_i)
 *     [nuu + nui, groups)      the sorted unique negatives that are nobody's positive in this batch (count: h_n_uniq_n[b])
 * h_seg_rows[b 3B + g] = the group's TABLE ROW (id + user_row0 for users, id + item_row0 for items: pass 0, n_users when
 * the items follow the users in one table; -1 for g >= groups), and group g owns entries [h_seg_end[b 3B + g - 1] (0 for
 * g = 0), h_seg_end[b 3B + g]) of the entry arrays at b 3B, slots ascending inside a group:
 *     h_seg[e]   = 4 * slot + role   (role 0: the slot's user, 1: its positive item, 2: its negative item)
 *     h_seg_a[e] = the table row the term needs: the slot's positive item (role 0), its user (roles 1, 2)
 *     h_seg_b[b B + e] = the slot's negative item's table row (role 0 entries: they are the first cnt entries of a batch)
 * Sizes: h_n_uniq_n nb; h_seg_rows / h_seg_end / h_seg / h_seg_a nb * 3 B; h_seg_b nb * B   (nb = ceil(n_edges / B)). */
srh_status_t srh_sampler_epoch_segments(srh_sampler_t* s, int64_t batch_size, const int32_t* h_u, const int32_t* h_i,
                                        const int32_t* h_j, const int32_t* h_uniq_u, const int32_t* h_n_uniq_u,
                                        const int32_t* h_uniq_i, const int32_t* h_n_uniq_i, int32_t user_row0,
                                        int32_t item_row0, int32_t* h_n_uniq_n, int32_t* h_seg_rows, int32_t* h_seg_end,
                                        int32_t* h_seg, int32_t* h_seg_a, int32_t* h_seg_b);
/* random.sample(range(n), k) (data/augmentor.py:35 edge_dropout keep-set; :15-16 node
 * dropout) replayed on the sampler's MT stream.  h_out receives k indices in draw order. */
srh_status_t srh_sampler_sample_range(srh_sampler_t* s, int64_t n, int64_t k, int64_t* h_out);
/* random.getrandbits(32) -- lets tests pin how much of the stream was consumed. */
srh_status_t srh_sampler_next_u32(srh_sampler_t* s, uint32_t* out);

/* find_k_largest(K, candidates) of reference util/algorithm.py:144-156 on the host, with the reference's order among EQUAL
 * scores: python's heapq on (score, position) tuples restated step for step (heapify, heapreplace on a strictly larger score,
 * a stable descending sort of the heap array).  The device ranking orders ties (score desc, id asc); base/graph_recommender.py
 * redoes the rows whose best K + 1 scores hold a tie through this call.  *out_count = min(K, n) entries written. */
srh_status_t srh_find_k_largest_host(int64_t k, const float* h_candidates, int64_t n, int64_t* h_out_ids,
                                     float* h_out_scores, int64_t* out_count);

/* `torch.rand(n)` of the CPU generator (ATen: mt19937, one 32-bit word per float32, value = (word & 0xFFFFFF) * 2^-24),
 * replayed on the host from the generator's own words: what model/graph/BUIR.py:118-121 draws per forward pass for its
 * sparse dropout -- `torch.floor(keep_prob + torch.rand(nnz)).type(torch.bool)` -- at 5 M entries per step on the Yelp2018
 * shape, where ATen's serial kernel takes 15 ns per draw.  h_mt624 / *pos are the generator's 624 state words and the
 * index of the next unread word (624: regenerate first), in and out.  h_out (nullable) receives the n uniforms; h_keep
 * (nullable) receives floorf(keep_addend + u) != 0 as bytes, the fp32 arithmetic of the torch expression. */
srh_status_t srh_mt19937_uniform_f32(uint32_t* h_mt624, int32_t* pos, int64_t n, float* h_out,
                                     float keep_addend, uint8_t* h_keep);

/* ------------------------------------------------------------------------------------
 * (a-2) Graph normalisation -- replaces data/graph.py:10-24 normalize_graph_mat and
 * data/ui_graph.py:58-65 convert_to_laplacian_mat on a device-resident CSR.
 *
 * The (N x N) bipartite adjacency keeps ONE structure for the graph and all its
 * edge-dropped views; a view is a value array.  d_edge_id[p] is the position, in the
 * row-major order of the U x I interaction matrix, of the interaction behind non-zero p
 * (both the (u,i) and the (i,u) copies carry the same id), d_weight[p] its weight (NULL =
 * all ones), d_keep[e] != 0 keeps interaction e (NULL = keep all).
 * Output: d_vals[p] = keep ? (dinv[row] * w) * dinv[col] : 0 with
 * dinv[r] = (sum of kept weights in row r)^-1/2, 0 for an empty row -- the reference's
 * inf -> 0 rule (graph.py:15).  d_deg_ws: workspace of n_rows floats.
 * Row-sharded graphs (the CSR holds the rows [row_offset, row_offset + n_rows) of the node table and
 * its columns index the whole table): d_deg_ws covers the whole table, phase 1 only fills this
 * shard's dinv entries, the caller all-gathers d_deg_ws, phase 2 only writes the values.
 * phase 0 = both in one call (row_offset 0 for a whole graph).
 * d_inv_sqrt_table (optional, table_len entries): table[k] = float32(k)^-1/2 as the HOST's
 * numpy computes it; integral degrees below table_len are looked up there, which makes the
 * values bit-identical to the reference's np.power(rowsum, -0.5) (numpy's fp32 pow is not
 * correctly rounded, so no device formula can match it in every last bit).
 * ---------------------------------------------------------------------------------- */
srh_status_t srh_adj_sym_normalize(int64_t n_rows, const int32_t* d_indptr,
                                   const int32_t* d_indices, const int32_t* d_edge_id,
                                   const float* d_weight, const uint8_t* d_keep,
                                   const float* d_inv_sqrt_table, int32_t table_len,
                                   float* d_deg_ws, float* d_vals, int64_t row_offset, int32_t phase,
                                   void* stream);

/* ------------------------------------------------------------------------------------
 * (a-3, a-4) Sparse propagation -- replaces torch.sparse.mm(adj, dense) at
 * LightGCN.py:72, XSimGCL.py:88, SimGCL.py:85, SGL.py:104-108 (adj produced by
 * base/torch_interface.py:8-13), with the per-layer elementwise tail fused in:
 *   PERTURB : y += sign(y) * normalize(noise_row) * eps        XSimGCL.py:90-91
 *   MEAN    : mean_out = (sum_t prev[t] + y) / mean_div         XSimGCL.py:95-96, LightGCN.py:74-75
 *   AXPY    : y = alpha * y + sum_t add_scale[t] * add[t]       (backward accumulation; add[t] may alias y)
 *   FANOUT  : with PERTURB, the SAME product also leaves differently-perturbed copies     SimGCL.py:81-93
 *             (SimGCL's clean pass and its two perturbed views multiply the same A by the same
 *             ego table in their first layer: one gather pass, three outputs)
 *
 * The plan is the row schedule (degree-sorted rows, heavy rows split into segments); it
 * depends only on indptr and is shared by every value array over the same structure.
 * ---------------------------------------------------------------------------------- */
typedef struct srh_spmm_plan srh_spmm_plan_t;

/* xcd_split_row: 0, or the first row of the second node class of a bipartite adjacency (= number
 * of users): rows below it are issued to XCDs 0-3, the rest to XCDs 4-7 (L2 locality only).
 * h_row_mid: NULL, or per row: m >= 0 -- the caller ordered the row's entries [column class 0 | column
 * class 1] and class 1 starts at entry m: the two parts are scheduled as separate segments on different
 * XCDs (each L2 then caches one column class of one row class) and summed in-kernel; -1 - c -- the row
 * stays whole and runs with column class c.  Locality only: the product is the same. */
srh_status_t srh_spmm_plan_create(srh_spmm_plan_t** out, int64_t n_rows, int64_t n_cols,
                                  const int32_t* h_indptr, int32_t split_len /* 0 = default */,
                                  int64_t xcd_split_row, const int32_t* h_row_mid);

/* Unequal XCD shares of a plan's task list (tables of d = 64 / 128 / 256 columns).  A task list binds task k to
 * workgroup k / 4 and the hardware runs workgroup b on XCD b % 8 (observed, performance only), so every XCD gets the same
 * NUMBER of workgroups -- but not the same time over them: at the Yelp2018 shape the XCDs finish a propagation launch
 * 3.5 us apart.  h_blocks_per_xcd[8] (>= 0, adding up to ceil(srh_spmm_plan_run_tasks() / 4) of the canonical list)
 * says how many workgroups of real tasks each XCD runs; queues above their share give up their last workgroups (the
 * shortest rows), the others append them, and empty records pad the list.  NULL restores the canonical list.  Same
 * tasks, same sums bit for bit; locality / balance only.  Synchronises the device: not inside a stream capture, and a
 * captured launch of this plan must be re-captured afterwards.  The engine calibrates the shares once at start-up
 * from srh_spmm_f32_probe (engine.py). */
srh_status_t srh_spmm_plan_set_xcd_shares(srh_spmm_plan_t* plan, int32_t d, const int32_t* h_blocks_per_xcd);
/* records in the list a launch on d-column tables runs now (a multiple of 32 once shares are set); -1: no such list */
int32_t srh_spmm_plan_run_tasks(const srh_spmm_plan_t* plan, int32_t d);
void srh_spmm_plan_destroy(srh_spmm_plan_t* plan);

enum { SRH_EPI_PERTURB = 1, SRH_EPI_MEAN = 2, SRH_EPI_AXPY = 4, SRH_EPI_ADAM = 8 };
#define SRH_MAX_ADAM_CLEAR 4
#define SRH_MAX_PREV 8
#define SRH_MAX_ADD 2
#define SRH_MAX_EXTRA 2

typedef struct srh_spmm_epilogue {
  int32_t flags;               /* OR of SRH_EPI_* ; 0 = plain y = A x                      */
  float eps;                   /* PERTURB magnitude                                          */
  const float* d_noise;        /* PERTURB: U[0,1) noise (n_rows, d) to inject, or NULL       */
  uint64_t rng_seed;        /* PERTURB with d_noise == NULL: in-kernel counter RNG,           */
  uint64_t rng_offset;      /*   counter = (rng_offset + *d_rng_step * rng_stride */
  const int64_t* d_rng_step;/*              + row, lane-quad); d_rng_step may be NULL  */
  uint64_t rng_stride;
  int32_t n_prev;              /* MEAN: number of earlier layer tensors                      */
  int32_t n_add;               /* AXPY: number of addends                                    */
  const float* d_prev[SRH_MAX_PREV];
  float mean_div;              /* MEAN: divisor (number of tensors averaged)                 */
  float alpha;                 /* AXPY: scale of the product (use 1 for a plain add)         */
  float* d_mean_out;           /* MEAN: (n_rows, d) output                                   */
  const float* d_add[SRH_MAX_ADD];
  float add_scale[SRH_MAX_ADD];
  /* Activity marks (any epilogue): a row / column is live iff mark[i] == (int32)*d_mark_stamp.
   * d_row_mark: rows that are not live are skipped entirely (y, mean_out keep their old
   * content).  d_col_mark: entries whose column is not live are treated as zero (the caller
   * guarantees x is zero there).  NULL = everything live. */
  const int32_t* d_row_mark;
  const int32_t* d_col_mark;
  const int64_t* d_mark_stamp;
  /* AXPY: bit t of add_sparse_mask says addend t is zero on every row whose d_add_mark entry is not
   * live, so it is only read on live rows (the batch-row gradients gF / gCL of the backward chain). */
  const int32_t* d_add_mark;
  int32_t add_sparse_mask;
  /* FANOUT (PERTURB only, no MEAN): n_extra further outputs extra_out[k] = perturb_k(product), each
   * with its own injected noise (or NULL) / counter offset; main_clean != 0 leaves the main output y
   * unperturbed. */
  int32_t n_extra;
  int32_t main_clean;
  float* d_extra_out[SRH_MAX_EXTRA];
  const float* d_extra_noise[SRH_MAX_EXTRA];
  uint64_t extra_rng_offset[SRH_MAX_EXTRA];
  /* Column-sharded tables (multi-GPU): y / x / every epilogue tensor hold columns
   * [noise_col0, noise_col0 + d) of rows that are noise_d_full wide on the whole job (any supported d).
   * Only PERTURB cares: its unit vector is normalised over the WHOLE row (XSimGCL.py:90), so d_noise /
   * d_extra_noise are (n_rows, noise_d_full) and the counter RNG regenerates the other ranks' columns.
   * 0 = the rows are d wide (everything above). */
  int32_t noise_d_full;
  int32_t noise_col0;
  /* Row scaling (any epilogue; d = 64 / 128 / 256): with r = d_row_scale[row] (e.g. D^-1/2 of the adjacency),
   *   SRH_SCALE_IN    the product is multiplied by r before everything else  -- with d_vals == NULL (pattern matrix,
   *                   every stored entry = 1) and x = D^-1/2 x' this IS D^-1/2 A D^-1/2 x' (graph.py:10-24) without
   *                   streaming a value per entry;
   *   SRH_SCALE_OUT   y (and the FANOUT outputs) are STORED multiplied by r -- the next layer's pre-scaled input
   *                   (mean_out is not: it holds true values);
   *   bit t of prev_unscale_mask / add_rowscale_mask: MEAN's d_prev[t] is stored pre-scaled (multiply by 1 / r,
   *                   0 where r = 0) / AXPY's d_add[t] is multiplied by r.
   * Lets the engine keep layer outputs in the pre-scaled domain between products. */
  const float* d_row_scale;
  int32_t scale_flags;
  int32_t prev_unscale_mask;
  int32_t add_rowscale_mask;
  /* Any `embedding.size` (base/recommender.py:16): tables of d_valid columns are stored zero-padded to the next width
   * the kernels serve.  Zero columns stay zero through every product, AXPY and MEAN; PERTURB must only know that the
   * noise row (torch.rand_like(h) in XSimGCL.py:90: d_valid columns) ends at noise_d_valid, so that F.normalize runs
   * over those columns alone.  0 = every column of the (whole) row is valid.  Tables of >= 64 columns. */
  int32_t noise_d_valid;
  /* ADAM (d = 64 / 128 / 256, no PERTURB / MEAN / SCALE_OUT, no row marks; the launch is the LAST product of a backward
   * chain, torch.optim.Adam(...).step() of XSimGCL.py:25,37 folded into it): the finished row -- the product after
   * SCALE_IN and AXPY -- is d loss / d param[row].  It is NOT stored (d_y is not written); instead row `row` of
   * d_adam_param / d_adam_m / d_adam_v takes srh_adam_step's update, operation for operation, with this step's
   * {lr / (1 - beta1^t), sqrt(1 - beta2^t)} read from d_adam_coef (float[2]: srh_batch_fetch writes them, see
   * d_adam_coef there).  As in srh_adam_step_reset: rows whose d_adam_clear_mark entry equals (int32)*d_mark_stamp are
   * zeroed in the adam_n_clear tables d_adam_clear[] ((n_rows, d); they may be this launch's addends, they must not be
   * its x), and d_adam_cursor (int64[2], or NULL) is advanced by one batch / one step -- d_mark_stamp must then point at
   * a COPY of the step (srh_batch_fetch's d_now), since the cursor moves while the launch runs. */
  float* d_adam_param;
  float* d_adam_m;
  float* d_adam_v;
  const float* d_adam_coef;
  float adam_beta1, adam_beta2, adam_eps;
  int32_t adam_n_clear;
  const int32_t* d_adam_clear_mark;
  float* d_adam_clear[SRH_MAX_ADAM_CLEAR];
  int64_t* d_adam_cursor;
} srh_spmm_epilogue_t;
enum { SRH_SCALE_IN = 1, SRH_SCALE_OUT = 2 };

/* y (n_rows, d) = A (CSR, fp32 values, int32 structure) * x (n_cols, d); d in {8,16,32,64,128,256}
 * (8 and 16: the thin-table kernel of the column-sharded layout).
 * x and y must not alias.  epi may be NULL.  d_vals may be NULL for d >= 64: A is then the PATTERN of the structure
 * (all stored entries 1) -- no value stream; column marks are not combined with it. */
srh_status_t srh_spmm_f32(const srh_spmm_plan_t* plan, const int32_t* d_indptr,
                          const int32_t* d_indices, const float* d_vals, const float* d_x,
                          float* d_y, int32_t d, const srh_spmm_epilogue_t* epi, void* stream);

/* y_v = A_v x for three value arrays over ONE structure in one traversal (the x rows are gathered once):
 * the first layer of SGL.py:104-108, where the full adjacency and its two edge-dropped views
 * (data/augmentor.py:29-40: same indptr / indices, dropped entries = 0) multiply the same ego table.
 * Bit-identical to three srh_spmm_f32 calls.  d = 64 only (SRH_ERR_UNSUPPORTED otherwise). */
srh_status_t srh_spmm3_f32(const srh_spmm_plan_t* plan, const int32_t* d_indices, const float* d_vals0,
                           const float* d_vals1, const float* d_vals2, const float* d_x, float* d_y0,
                           float* d_y1, float* d_y2, int32_t d, void* stream);

/* ------------------------------------------------------------------------------------
 * (a-5, a-6, a-7) Row gather + BPR + L2 regulariser, forward and backward in one call --
 * replaces  emb[idx]  (XSimGCL.py:30), util/loss_torch.py:6-10 bpr_loss and :18-22
 * l2_reg_loss, and their autograd.
 *
 * Batch rows b = 0..B-1:  u = U[u_idx[b]], p = I[i_idx[b]], n = I[j_idx[b]].
 *   bpr  = mean_b( -log(1e-5 + sigmoid(<u,p> - <u,n>)) )
 *   reg  = reg_coef * ( ||Ru||_F + ||Rp||_F [+ ||Rn||_F] ) / B_rows     (Frobenius, not squared)
 * where R* are the same gathered rows of the REG source tables (d_reg_user/d_reg_item;
 * pass the propagated tables for XSimGCL/SimGCL/SGL/MF, the ego tables for LightGCN,
 * LightGCN.py:25).  reg_coef already contains any extra /batch_size factor.
 * Gradients (d loss / d table rows, scaled by loss_scale) are ATOMICALLY ACCUMULATED into
 * d_g_user / d_g_item (the gradient w.r.t. d_user/d_item) and d_greg_user / d_greg_item
 * (w.r.t. the REG tables; may alias d_g_*).  d_losses[0] += bpr, d_losses[1] += reg
 * (double, caller zeroes).  d_n_rows: optional device int32 overriding B (graph replay).
 * d_ws: workspace of at least srh_bpr_ws_bytes(B) bytes.
 * (Float atomics make the low bits of a row that several pairs name depend on arrival order.  The struct forms below --
 * srh_bpr_l2_fwd_bwd_p, srh_bpr_infonce_fwd_bwd -- take the batch's row -> slot lists (srh_batch_segments_t) and sum
 * every row's contributions in slot order instead: bit-reproducible, no atomics.)
 * ---------------------------------------------------------------------------------- */
/* The batch's row groups and their slot lists on the DEVICE (srh_sampler_epoch_segments' arrays, uploaded).
 * d_batch_no == NULL: the arrays are this batch's slices and d_n_uniq_n its count.  d_batch_no != NULL (graph replay):
 * they are the EPOCH arrays and batch b = *d_batch_no lives at + b 3B (d_seg_b: + b B), d_n_uniq_n[b] (B = the problem's B). */
typedef struct srh_batch_segments {
  const int32_t* d_n_uniq_u;  /* device-side counts of the batch's unique users / positive items (srh_batch_fetch's d_meta[1], [2]) */
  const int32_t* d_n_uniq_i;
  const int32_t* d_n_uniq_n;
  const int32_t* d_seg_rows;
  const int32_t* d_seg_end;
  const int32_t* d_seg;
  const int32_t* d_seg_a;
  const int32_t* d_seg_b;
  const int32_t* d_batch_no;
  int32_t nce_rows;           /* how the InfoNCE problems of the same call name these rows (srh_bpr_infonce_fwd_bwd):
                                 0 none; 1: problem 0's row i is user group i, problem 1's row i is positive-item group i
                                 (XSimGCL.py:46-49, SimGCL.py:44-49); 2: problem 0's rows are [users ; positive items]
                                 (SGL.py:120-125).  Rows of a problem are then finished by the group that owns them. */
  int32_t rows_are_zero;      /* != 0: the caller guarantees that every row this batch names is ZERO in every gradient table
                                 of the call (the engine clears exactly those rows after every step): rows are stored, not
                                 read-added-stored -- one memory round trip less in a latency-bound launch */
} srh_batch_segments_t;
int64_t srh_bpr_ws_bytes(int64_t B);
srh_status_t srh_bpr_l2_fwd_bwd(const float* d_user, const float* d_item,
                                const float* d_reg_user, const float* d_reg_item,
                                const int32_t* d_u_idx, const int32_t* d_i_idx,
                                const int32_t* d_j_idx, int64_t B, const int32_t* d_n_rows,
                                int32_t d, float reg_coef, int32_t reg_include_neg,
                                float loss_scale, float* d_g_user, float* d_g_item,
                                float* d_greg_user, float* d_greg_item, double* d_losses,
                                void* d_ws, void* stream);

/* Plain (non-gathered) forms behind the op-level tier's loss functions (selfrec_amd/util/loss_torch.py: the signatures of
 * util/loss_torch.py:6-22).  Each call is ONE launch and leaves its scalar on the device as f32, the way the reference's
 * expression would: the last workgroup to finish (a ticket in d_scalar_ws) turns the double accumulators into the result.
 * d_scalar_ws: SRH_SCALAR_WS_BYTES bytes, zero before its first use; every call leaves it zero again (calls that share
 * one must be ordered on a stream).  The upstream gradient d_gout is a DEVICE scalar: no host synchronisation in a
 * backward pass. */
#define SRH_SCALAR_WS_BYTES 64
/* bpr_loss (loss_torch.py:6-10):  *d_loss = mean_b -log(1e-5 + sigmoid(u_b.p_b - u_b.n_b));  d_coef[b] = d loss / d (pos_b - neg_b) */
srh_status_t srh_bpr_fwd(const float* d_u, const float* d_p, const float* d_n, int64_t B,
                         int32_t d, void* d_scalar_ws, float* d_loss, float* d_coef /* B */,
                         void* stream);
srh_status_t srh_bpr_bwd(const float* d_u, const float* d_p, const float* d_n,
                         const float* d_coef, int64_t B, int32_t d, const float* d_gout,
                         float* d_gu, float* d_gp, float* d_gn, void* stream);
/* l2_reg_loss(reg, *embs) (loss_torch.py:18-22):  *d_loss = reg * sum_k ||x_k||_F / rows_k, summed left to right in f32;
 * d_norms[k] = ||x_k||_F (kept for the backward pass).  Backward: d_gx_k = x_k * (((*d_gout * reg) / rows_k) / ||x_k||),
 * zero where the norm is zero (torch.norm's subgradient).  1..4 blocks of rows per call; `blocks` is a HOST array. */
typedef struct srh_l2_block {
  const float* d_x; /* (rows, cols) contiguous f32 */
  int64_t rows;
  int64_t cols;
  float* d_gx;      /* backward only */
} srh_l2_block_t;
srh_status_t srh_l2_reg_fwd(const srh_l2_block_t* blocks, int32_t n_blocks, float reg, void* d_scalar_ws,
                            float* d_norms /* n_blocks */, float* d_loss, void* stream);
srh_status_t srh_l2_reg_bwd(const srh_l2_block_t* blocks, int32_t n_blocks, float reg, const float* d_norms,
                            const float* d_gout, void* stream);

/* ------------------------------------------------------------------------------------
 * (a-8) InfoNCE forward + backward -- replaces util/loss_torch.py:35-50 InfoNCE(view1,
 * view2, temperature, b_cos=True) at XSimGCL.py:48-49, SimGCL.py:48-49, SGL.py:125.
 *
 *   v1 = normalize(V1[idx]), v2 = normalize(V2[idx])   (rows gathered when d_idx != NULL)
 *   S = v1 v2^T / tau ;  loss = -mean_i log_softmax(S, dim=1)[i,i]
 * S (n x n) is never materialised: one flash-style pass over row tiles produces the row
 * log-sum-exp and dL/dv1, a second pass over column tiles produces dL/dv2; the
 * normalisation backward and the scatter to the source rows are fused in the last pass.
 * d_loss[0] += loss_scale * loss (double).  Gradients scaled by loss_scale are ADDED to
 * rows idx of d_g1 / d_g2 (row i when d_idx == NULL; idx must then be unique per call).
 * d_n: optional device int32 overriding n.  d_ws >= srh_infonce_ws_bytes(n, d) bytes.
 * ---------------------------------------------------------------------------------- */
int64_t srh_infonce_ws_bytes(int64_t n, int32_t d);
srh_status_t srh_infonce_fwd_bwd(const float* d_v1, const float* d_v2, const int32_t* d_idx,
                                 int64_t n, const int32_t* d_n, int32_t d, float tau,
                                 float loss_scale, double* d_loss, float* d_g1, float* d_g2,
                                 void* d_ws, void* stream);

/* Several InfoNCE problems in one set of launches (XSimGCL's user side and item side share every
 * kernel: half the launches, twice the resident waves).  d_ws must hold the SUM of
 * srh_infonce_ws_bytes(problems[k].n, d); at most 4 problems per call.  `problems` is a HOST array. */
typedef struct srh_infonce_problem {
  const float* d_v1;
  const float* d_v2;
  const int32_t* d_idx;
  int64_t n;
  const int32_t* d_n;
  float* d_g1;
  float* d_g2;
  int32_t g2_exclusive; /* != 0: the rows of d_g2 this problem names are written by nobody else during the call (e.g. the
                           contrast-layer gradient table of XSimGCL: user-side and item-side problems name disjoint rows):
                           plain read-add-store instead of 4 device-scope atomics per lane */
} srh_infonce_problem_t;
/* precision: SRH_NCE_SPLIT16 / SRH_NCE_F32 for THIS call, or SRH_NCE_DEFAULT (the process default below). */
srh_status_t srh_infonce_fwd_bwd_multi(const srh_infonce_problem_t* problems, int32_t n_problems,
                                       int32_t d, float tau, float loss_scale, double* d_loss,
                                       void* d_ws, int32_t precision, void* stream);

/* Arithmetic of InfoNCE's two n x n x d products (util/loss_torch.py:46-47's matmul and its backward).  The reference
 * computes them in fp32, and so does the DEFAULT here (SRH_NCE_F32): every multiply-add on v_mfma_f32_16x16x4_f32 (exact
 * f32 fma chains), d = 64 / 128 (d = 256 resolves to the split mode).  SRH_NCE_SPLIT16 is the faster opt-in: operands
 * carried as short sums of 16-bit pieces on the 16-bit MFMA pipe with f32 accumulation -- the similarity product on scaled
 * f16 hi + lo (x to 2^-22: logits as accurate as an f32 dot product), the P.V product on bf16 hi + mid (2^-18 per product;
 * gradients within 1e-6 relative of the f64 expression).  The mode is an ARGUMENT of srh_infonce_fwd_bwd_multi /
 * srh_bpr_infonce_fwd_bwd (a trainer carries its own: two models in one process do not share it);
 * srh_infonce_set_precision sets the process DEFAULT that SRH_NCE_DEFAULT and srh_infonce_fwd_bwd resolve to, and the
 * environment variable SRH_NCE_SPLIT16 (set to anything) selects the split mode as its initial value.  (SRH_NCE_SPLIT_BF16:
 * the mode's name in ABI <= 20.) */
#define SRH_NCE_SPLIT16 0
#define SRH_NCE_SPLIT_BF16 SRH_NCE_SPLIT16
#define SRH_NCE_F32 1
#define SRH_NCE_DEFAULT (-1)
srh_status_t srh_infonce_set_precision(int32_t mode);
int32_t srh_infonce_get_precision(void);

/* The whole batch objective of XSimGCL.py:30-35 / SimGCL.py:33-36 / SGL.py:33-37 --
 *   rec_loss + l2_reg_loss + cl_rate * cl_loss  and its gradients w.r.t. the gathered tables --
 * in one call: exactly srh_bpr_l2_fwd_bwd(bpr...) followed by srh_infonce_fwd_bwd_multi(problems...),
 * with the same outputs, but the two losses' O(batch) kernels share launches (4 instead of 6).
 * `bpr` mirrors the argument list of srh_bpr_l2_fwd_bwd; both structs are HOST memory. */
typedef struct srh_bpr_problem {
  const float* d_user;
  const float* d_item;
  const float* d_reg_user;
  const float* d_reg_item;
  const int32_t* d_u_idx;
  const int32_t* d_i_idx;
  const int32_t* d_j_idx;
  int64_t B;
  const int32_t* d_n_rows;
  float reg_coef;
  int32_t reg_include_neg;
  float loss_scale;
  float* d_g_user;
  float* d_g_item;
  float* d_greg_user;
  float* d_greg_item;
  double* d_losses;
  void* d_ws; /* srh_bpr_ws_bytes(B) */
  const srh_batch_segments_t* seg; /* HOST pointer or NULL.  Given: every touched row of the gradient tables is written by the
                                      ONE row group that owns it -- read, add the row's contributions in slot order (and the
                                      InfoNCE gradients of that row: nce_rows), store -- no float atomics: the same bits on
                                      every run.  The lists hold TABLE ROWS: user rows index d_user / d_reg_user / d_g_user /
                                      d_greg_user, item rows the item tables (two tables with ids, or one table passed for
                                      both with the items offset: user_row0 / item_row0 of srh_sampler_epoch_segments); an
                                      InfoNCE problem's d_idx must name the same rows of ITS tables.  User rows and item rows
                                      must not alias.  NULL: atomic accumulation as srh_bpr_l2_fwd_bwd describes. */
} srh_bpr_problem_t;
/* srh_bpr_l2_fwd_bwd with its arguments in the struct (and the optional fixed-order reduction of `seg`). */
srh_status_t srh_bpr_l2_fwd_bwd_p(const srh_bpr_problem_t* bpr, int32_t d, void* stream);
srh_status_t srh_bpr_infonce_fwd_bwd(const srh_bpr_problem_t* bpr, const srh_infonce_problem_t* problems,
                                     int32_t n_problems, int32_t d, float tau, float cl_scale,
                                     double* d_cl_loss, void* d_nce_ws, int32_t precision, void* stream);

/* ------------------------------------------------------------------------------------
 * (a-9) Dense Adam -- replaces torch.optim.Adam(...).step() at XSimGCL.py:25,37
 * (betas, eps as given; no weight decay; bias-corrected).  `step` is 1-based; when
 * d_step != NULL the step count is read from that device int64 (graph replay).
 * ---------------------------------------------------------------------------------- */
srh_status_t srh_adam_step(float* d_param, const float* d_grad, float* d_m, float* d_v,
                           int64_t n_elem, int64_t step, const int64_t* d_step, float lr,
                           float beta1, float beta2, float eps, void* stream);
/* The same update of an (n_rows, d) table as the LAST kernel of the engine's step: rows whose activity mark
 * d_row_mark[row] equals *d_step are zeroed in up to SRH_MAX_ADAM_CLEAR other (n_rows, d) tables in the same pass
 * (the batch-sparse gradient buffers: optimizer.zero_grad() of XSimGCL.py:35 for the only rows that are non-zero),
 * and d_cursor_advance (int64[2], optional) is incremented.  d_step must then be srh_batch_fetch's d_now copy, not
 * the cursor itself.  A table to clear may be d_grad (MF: the gradient buffer is the batch-sparse one). */
srh_status_t srh_adam_step_reset(float* d_param, const float* d_grad, float* d_m, float* d_v, int64_t n_rows,
                                 int32_t d, const int64_t* d_step, float lr, float beta1, float beta2, float eps,
                                 const int32_t* d_row_mark, int32_t n_clear, float* const* d_clear_tables,
                                 int64_t* d_cursor_advance, void* stream);

/* ------------------------------------------------------------------------------------
 * (a-10, a-11) Full-catalogue scoring + training-item mask + top-K -- replaces the per
 * user loop of base/graph_recommender.py:46-53 (predict = XSimGCL.py:57-60, mask = -10e8,
 * util/algorithm.py:144-156 find_k_largest).
 *
 *   scores[r, :] = item_emb @ user_emb[user_ids[r]]       fp32 MFMA (exact-f32 fma chain)
 *   scores[r, i] = -1e9 for i in training items of that user (CSR d_r_indptr/d_r_indices)
 *   top-K by (score desc, id asc), written best-first.
 * d_scores_ws: (ws_rows, n_items) fp32 scratch; the queries pass through it ws_rows at a time (size
 * it to stay cache-resident: <= ~96 MB).  d_user_ids may be NULL (rows 0..n_query-1 of d_user_emb).
 * ---------------------------------------------------------------------------------- */
srh_status_t srh_score_mask_topk(const float* d_user_emb, const int32_t* d_user_ids,
                                 int64_t n_query, const float* d_item_emb, int64_t n_items,
                                 int32_t d, const int32_t* d_r_indptr, const int32_t* d_r_indices,
                                 int32_t k, float* d_scores_ws, int64_t ws_rows, int32_t* d_out_ids,
                                 float* d_out_scores, void* stream);
/* The same ranking without ever storing the (users x items) scores.  Per chunk of chunk_rows users:
 *   1. the exact pipeline above on the first `sample_items` items only; its K-th score t_u is a lower
 *      bound of user u's overall K-th score;
 *   2. the scoring GEMM over the whole catalogue with a filter epilogue: a score is kept iff it is
 *      >= t_u -- a few hundred (item, score) pairs per user, appended to a list of `cap` slots;
 *   3. training items of u are dropped from its list (binary search in the mask CSR) and the rest is
 *      put in exact (score desc, id asc) order.
 *   d = 64 / 128: passes 1 and 2 only have to DECIDE, so they run on bf16 operands (one v_mfma_f32_32x32x16_bf16 per 16
 *   dimensions: 16x the f32 MFMA's rate) against rigorous PER-ITEM error bounds: |s~_uj - s_uj| <= delta_uj = 3.94e-3 |u| |i_j|
 *   (both operands rounded to bf16).  Pass 1 ranks LOWER bounds s~ - delta of the sample's scores, so t_u is a lower bound of
 *   the exact K-th best; pass 2 keeps (id, s~) with s~ + delta >= t_u; pass 3 takes tau = the K-th largest s~ - delta of the
 *   unmasked survivors and re-scores exactly -- by the scalar fma chain that is bit-identical to the f32 MFMA's accumulation --
 *   the survivors with s~ + delta >= tau.  (Round 3: three split-bf16 products and one margin 4e-5 |u| max_j|i_j| per user.)
 *   In this path the bound slice of pass 1 is the `sample_items` items of LARGEST NORM, not the first ones: the item image is
 *   built in norm order (norms, one radix sort of (norm, id) pairs, the permutation and its inverse), a user's best scores sit
 *   on the items training has pushed outwards, and the K-th best over such a slice is a far tighter bound: 39 instead of 60
 *   survivors per user on trained tables (tools/sample_choice_probe.py).
 * Scores come from the same fma chain, so ids and scores are identical to srh_score_mask_topk.
 * d_out_counts[q] = number of survivors of row q, training items included: when it exceeds `cap`
 * (tie-heavy rows, users with thousands of training items) that row of the outputs is NOT valid and the caller ranks it with
 * srh_score_mask_topk.  d_ws: srh_score_mask_topk_filtered_ws_bytes(chunk_rows, ...) bytes.
 * Shape constants measured on the Yelp2018 shape (profiles/r03_m_*): chunk_rows 16384 (larger chunks fill the chip: 4096 ->
 * 16384 users per chunk is +20 %), sample_items 3072 in norm order (2048 .. 4096 are within 3 % of each other on trained
 * tables; round 3, catalogue order: 4096), cap 1024.  In the split path, pass 1 derives t~_u as the K-th largest of 256 disjoint group maxima of
 * the masked sample scores (a lower bound of their K-th largest: K distinct items attain it). */
int64_t srh_score_mask_topk_filtered_ws_bytes(int64_t chunk_rows, int64_t sample_items, int32_t k, int32_t cap,
                                              int64_t n_items, int32_t d);
srh_status_t srh_score_mask_topk_filtered(const float* d_user_emb, const int32_t* d_user_ids, int64_t n_query,
                                          const float* d_item_emb, int64_t n_items, int32_t d,
                                          const int32_t* d_r_indptr, const int32_t* d_r_indices, int32_t k,
                                          int64_t sample_items, int32_t cap, int64_t chunk_rows, void* d_ws,
                                          int32_t* d_out_ids, float* d_out_scores, int32_t* d_out_counts,
                                          void* stream);
/* (n_query, k1) ranked ids / scores (k1 = K + 1) -> their first K columns; a row in which two NEIGHBOURS of the K + 1 scores are
 * EQUAL is marked d_out_ids[row][0] = -1 - id: its order among the equal scores is the reference's heap walk's
 * (util/algorithm.py:144-156), which the caller restores on the host (srh_find_k_largest_host). */
srh_status_t srh_topk_trim_mark_ties(const int32_t* d_ids, const float* d_scores, int64_t n_query, int32_t k1,
                                     int32_t* d_out_ids, float* d_out_scores, void* stream);
/* The scoring GEMM alone: C (m, n) = A (m, d) B (n, d)^T, fp32 MFMA. */
srh_status_t srh_gemm_nt_f32(const float* d_a, const float* d_b, float* d_c, int64_t m,
                             int64_t n, int32_t d, void* stream);
/* Row-wise top-K of a (rows, n) fp32 matrix (ld = n). */
srh_status_t srh_topk_rows(const float* d_scores, int64_t rows, int64_t n, int32_t k,
                           int32_t* d_out_ids, float* d_out_scores, void* stream);

/* (f-3) Metric tail: d_flags[q*k + r] = 1 iff d_ids[q*k + r] is a test item of user d_user_ids[q]
 * (rows 0..n_query-1 when NULL), the membership test of util/evaluation.py:7-16 (hits) and :66-78
 * (NDCG).  d_t_indptr / d_t_indices: the test set as a (users x items) CSR with sorted columns. */
srh_status_t srh_topk_hit_flags(const int32_t* d_ids, int64_t n_query, int32_t k,
                                const int32_t* d_user_ids, const int32_t* d_t_indptr,
                                const int32_t* d_t_indices, uint8_t* d_flags, void* stream);
/* (f-3) Per-user figures of util/evaluation.py:7-16,66-78 from the hit flags, for up to 8 cut-offs N <= k at once:
 *   d_hits[c * n_query + q] = number of hits among the first cuts[c] ranked items of row q                (Metric.hits)
 *   d_ndcg[c * n_query + q] = DCG / IDCG of row q at cuts[c], accumulated exactly as Metric.NDCG does it: float64,
 *                             positions best first, gain table d_gains[pos] = 1 / log2(pos + 2) and ideal prefix sums
 *                             d_ideal[c * (k + 1) + min(|truth_q|, cuts[c])] as the HOST computed them (python's
 *                             math.log: passed in, not recomputed, so the quotients are the reference's bit for bit).
 * d_sizes[q] = |truth_q| (> 0).  cuts: host array.  The cross-user sums stay on the host (31.5 k adds). */
srh_status_t srh_metric_rows(const uint8_t* d_flags, const int32_t* d_sizes, int64_t n_query, int32_t k,
                             const int32_t* cuts, int32_t n_cuts, const double* d_gains, const double* d_ideal,
                             int32_t* d_hits, double* d_ndcg, void* stream);

/* ------------------------------------------------------------------------------------
 * Small device utilities used by the fused engine.
 * ---------------------------------------------------------------------------------- */
/* y = a*x + b*y   (elementwise, n_elem floats; y may be uninitialised when b == 0) */
srh_status_t srh_axpby(float a, const float* d_x, float b, float* d_y, int64_t n_elem, void* stream);
/* Batch cursor for graph replay.  d_cursor: int64[2] = {batch number within the epoch, optimiser
 * step (1-based)} in device memory.  srh_batch_fetch copies batch d_cursor[0] of the per-epoch index
 * arrays into fixed staging buffers and publishes its sizes (d_meta: int32[4] = {rows, n_uniq_u,
 * n_uniq_i, batch_no}); it only READS the cursor.  The cursor is advanced by the last kernel of the
 * step: srh_zero_rows(..., d_cursor_advance, ...) or srh_cursor_advance. */
typedef struct {
  const int32_t* d_epoch_u;        /* the epoch's sampled triples (srh_sampler_epoch, uploaded) */
  const int32_t* d_epoch_i;
  const int32_t* d_epoch_j;
  const int32_t* d_epoch_uniq_u;   /* optional: per-batch sorted unique ids + counts (all five together) */
  const int32_t* d_epoch_uniq_i;
  const int32_t* d_n_uniq_u;
  const int32_t* d_n_uniq_i;
  int64_t n_edges;
  int64_t batch_size;
  const int64_t* d_cursor;         /* [0] = batch no, [1] = adam step */
  int32_t* d_stage_u;
  int32_t* d_stage_i;
  int32_t* d_stage_j;
  int32_t* d_stage_uniq_u;
  int32_t* d_stage_uniq_i;
  int32_t* d_meta;
  int32_t* d_row_mark;             /* optional: mark[u] = mark[off+i] = mark[off+j] = step */
  int32_t mark_item_offset;
  int32_t cat_item_offset;
  double* d_zero4;                 /* optional: 4 loss accumulators cleared for the new step */
  int32_t* d_stage_cat;            /* optional (2*batch_size): [uniq users ; uniq items + cat_item_offset] */
  int32_t* d_n_cat;                /* optional: n_uniq_u + n_uniq_i */
  int64_t* d_now;                  /* optional int64[2]: copy of d_cursor taken by this launch -- what a kernel that runs
                                      CONCURRENTLY with the cursor advance (Adam beside srh_zero_rows) reads its step from */
  int64_t half_batches;            /* 0: the arrays hold one epoch.  nb > 0: they hold TWO epochs back to back, nb batch
                                      slots each (d_epoch_u/i/j and the unique-id lists: 2 nb batch_size entries, the counts:
                                      2 nb) -- batch no b >= nb is batch b - nb of the second: its rows are cut against
                                      n_edges with b - nb, its data sits at b.  The engine fills one half (a copy on its own
                                      stream) while the steps read the other: an epoch boundary costs a cursor write. */
  float* d_adam_coef;              /* optional float[2]: this step's Adam constants {adam_lr / (1 - adam_beta1^t),
                                      sqrt(1 - adam_beta2^t)}, t = d_cursor[1], computed in double as torch does -- what
                                      an SRH_EPI_ADAM product of the step reads (one pow per step, not one per wave) */
  float adam_lr, adam_beta1, adam_beta2;
} srh_batch_fetch_args_t;
srh_status_t srh_batch_fetch(const srh_batch_fetch_args_t* args, void* stream);
/* srh_spmm_f32 for d = 64, 128, 256 (no column marks) that ALSO performs srh_batch_fetch(fetch): eight more workgroups at
 * the head of the same launch.  For a step whose first product does not depend on the batch (every model here: the
 * batch only enters at the last forward layer's row marks and at the losses) this removes a 5 us launch from the
 * step's critical path.  The product must not read anything the fetch writes (row marks, staged ids). */
srh_status_t srh_spmm_f32_with_fetch(const srh_spmm_plan_t* plan, const int32_t* d_indptr, const int32_t* d_indices,
                                     const float* d_vals, const float* d_x, float* d_y, int32_t d,
                                     const srh_spmm_epilogue_t* epi, const srh_batch_fetch_args_t* fetch, void* stream);

/* srh_spmm_f32 (no column marks, d = 64 / 128 / 256) that also leaves, for task k of the list, d_stamps[3k .. 3k+2] =
 * {begin, end} of its wave on the chip-wide 100 MHz clock and the XCD (0 .. 7) it ran on; records of tasks that do not
 * exist in the list stay untouched.  d_stamps: 3 * srh_spmm_plan_run_tasks(plan, d) uint64 of device memory.  What the
 * calibration of srh_spmm_plan_set_xcd_shares reads; tools/spmm_lab/run.py --probe draws timelines from it. */
srh_status_t srh_spmm_f32_probe(const srh_spmm_plan_t* plan, const int32_t* d_indices, const float* d_vals,
                                const float* d_x, float* d_y, int32_t d, const srh_spmm_epilogue_t* epi,
                                uint64_t* d_stamps, void* stream);
/* The propagation launch's own lower bound, measured (bench.py's roofline.gather_bound_us): the task list of `plan` for d-
 * column tables on its workgroup -> XCD placement, the (col) stream and the eight-in-flight gathers of x rows with their
 * multiply-adds -- and nothing after them: no values (pattern), no cross-group reduction, no split-row hand-off, no epilogue,
 * no y (d_scratch: one (d)-float row, never written for finite sums).  A strict subset of srh_spmm_f32's work on its schedule:
 * it cannot come out slower than the product it bounds.  The XSimGCL_Encoder.forward product it prices:
 * model/graph/XSimGCL.py:83-101.  d = 64 / 128 / 256.  Measurement only. */
srh_status_t srh_spmm_gather_bound(const srh_spmm_plan_t* plan, const int32_t* d_indices, const float* d_x, float* d_scratch,
                                   int32_t d, void* stream);
/* The bare gather stream of a propagation launch, for the roofline block of bench.py (the step's dominant
 * kernel is priced against HBM on ALGORITHMIC bytes; this probe says what the chip's vector-memory path needs for the row
 * fetches alone).  Walks d_indices[0 .. n_idx) -- the live graph's CSR column array, as data/ui_graph.py:47-56 orders it --
 * and fetches row d_indices[k] of the (n_x_rows, d) fp32 table d_x for every k, 8 fetches in flight per row-group, summing
 * into registers: no values, no epilogue, no output (d_sink: 16 bytes, never written for finite tables).  `blocks`
 * workgroups of 256 threads.  d = 64 / 128 / 256.  Replaces nothing in the reference: measurement only. */
srh_status_t srh_gather_floor_probe(const int32_t* d_indices, int64_t n_idx, const float* d_x, int64_t n_x_rows, int32_t d,
                                    int32_t blocks, float* d_sink, void* stream);
/* Zero the listed rows of up to SRH_MAX_ZERO_LISTS (rows, d) tables in one launch: rows
 * d_idx[k][0 .. count_k) + row_offset[k] of d_tables[k], count_k = *d_counts[k] (or n_max[k] when
 * d_counts[k] is NULL).  The sparse counterpart of a memset for gradient buffers that only
 * ever receive O(batch) non-zero rows.  Array arguments are HOST arrays of device pointers. */
#define SRH_MAX_ZERO_LISTS 8
srh_status_t srh_zero_rows(int32_t n_lists, float* const* d_tables, const int32_t* const* d_idx,
                           const int32_t* const* d_counts, const int32_t* n_max,
                           const int32_t* row_offset, int32_t d,
                           int64_t* d_cursor_advance /* optional: {batch, step} += 1 */, void* stream);
srh_status_t srh_cursor_advance(int64_t* d_cursor, void* stream);

/* ------------------------------------------------------------------------------------
 * (e) Column-sharded tables -- the multi-GPU layout for graphs that fit one GPU (DESIGN.md
 * section 6).  Rank r keeps columns [r*dl, (r+1)*dl) of every (N, d) table, dl = d / G.  Propagation
 * (LightGCN.py:72, XSimGCL.py:88) is independent per column: no exchange.  The losses
 * (XSimGCL.py:30-33,45-50; loss_torch.py:6-10,18-22,35-50) read whole rows, but only O(batch) of
 * them: the rows the staged batch lists name.  Those are exchanged by ONE all-gather per step:
 *   srh_batch_pack -> all-gather (host: RCCL) -> srh_batch_unpack -> the loss kernels above on the
 *   "compact" (5B, d) tables with the constant index lists slot -> slot -> srh_batch_scatter.
 * Compact row k = slot k of [u | i | j | unique users | unique items] (B slots each; slots past a
 * list's device-side count are dead: never written, never read).
 * ------------------------------------------------------------------------------------ */
#define SRH_MAX_EXCHANGE 8
typedef struct srh_batch_lists {
  const int32_t* d_idx[5];     /* staged lists of srh_batch_fetch: u, i, j, uniq_u, uniq_i (table rows)  */
  const int32_t* d_count[5];   /* device-side live counts (d_meta[0] for u/i/j, [1], [2]); NULL = B      */
  int32_t B;                   /* slots per list                                                         */
} srh_batch_lists_t;
/* d_send (n_tables, 5B, dl): the listed rows of each local (N, dl) table (dead slots: zeros).
 * d_cat_idx / d_n_cat (optional, 2B / 1): compact slots of [unique users ; unique items] as one list
 * (SGL.py:120-125 concatenates the two sides). */
srh_status_t srh_batch_pack(const srh_batch_lists_t* lists, int32_t n_tables, const float* const* d_tables,
                            int32_t dl, float* d_send, int32_t* d_cat_idx, int32_t* d_n_cat, void* stream);
/* d_recv (world, n_tables, 5B, dl) = all-gather of every rank's d_send -> d_compact[t] (5B, world*dl);
 * the live rows of the n_grads compact gradient tables are zeroed. */
srh_status_t srh_batch_unpack(const srh_batch_lists_t* lists, int32_t n_tables, int32_t world, int32_t dl,
                              const float* d_recv, float* const* d_compact, int32_t n_grads,
                              float* const* d_compact_grads, void* stream);
/* d_local_grads[p][node, :] += d_compact_grads[p][slot, col0 : col0 + dl] for every live slot. */
srh_status_t srh_batch_scatter(const srh_batch_lists_t* lists, int32_t n_pairs, const float* const* d_compact_grads,
                               float* const* d_local_grads, int32_t d_full, int32_t col0, int32_t dl, void* stream);

/* ------------------------------------------------------------------------------------
 * (e) Row-sharded tables -- SURVEY 8e's partition: rank r owns n rows of every (N, d) table (and the CSR rows of its nodes),
 * tables are kept in all-gather order (row = owner * n + local row).  The reference has one implicit device
 * (model/graph/XSimGCL.py:24,73); these are the exchanges its encoder (XSimGCL.py:83-101, one torch.sparse.mm per layer)
 * needs once the rows are dealt over N > 1 GPUs:
 *   forward,  before layer k:  srh_allgather_rows      every rank's (n, d) slice of E^(k-1) -> the whole (world n, d) table
 *   backward, after  layer k:  srh_reducescatter_rows  every rank's partial sums over ALL rows -> the owners' (n, d) slices
 *   (A_hat is symmetric: a backward product written by owners needs only the all-gather; the reduce-scatter serves
 *   formulations that produce partial rows.)  srh_allreduce_sum_f32: the dense gradient of the data-parallel layout.
 * Thin wrappers over RCCL (ncclAllGather / ncclReduceScatter / ncclAllReduce on ncclFloat32), asynchronous on `stream`.
 * `comm` is an ncclComm_t passed as void*: made by srh_comm_init_rank, or the caller's own.  RCCL is not linked -- the
 * symbols are resolved among what the process has loaded (then librccl.so.1), so the caller's copy of RCCL is the one
 * used.  Without RCCL every call returns SRH_ERR_UNSUPPORTED with a message.  In-place is allowed where RCCL allows it
 * (d_rows == d_table + rank * n * d).
 * ---------------------------------------------------------------------------------- */
#define SRH_COMM_ID_BYTES 128
/* rank 0 makes the id (ncclGetUniqueId), hands it to the others by any host channel; then every rank joins. */
srh_status_t srh_comm_unique_id(uint8_t* h_id /* SRH_COMM_ID_BYTES */);
srh_status_t srh_comm_init_rank(void** out_comm, int32_t world, int32_t rank, const uint8_t* h_id);
srh_status_t srh_comm_destroy(void* comm);
srh_status_t srh_comm_world(void* comm, int32_t* out_world);
srh_status_t srh_allgather_rows(const float* d_rows /* (n_rows, d) */, float* d_table /* (world n_rows, d) */,
                                int64_t n_rows, int32_t d, void* comm, void* stream);
srh_status_t srh_reducescatter_rows(const float* d_table /* (world n_rows, d) */, float* d_rows /* (n_rows, d) */,
                                    int64_t n_rows, int32_t d, void* comm, void* stream);
srh_status_t srh_allreduce_sum_f32(float* d_buf, int64_t n_elem, void* comm, void* stream);

/* ------------------------------------------------------------------------------------
 * (f-1) Dataset files -> id arrays -- replaces the python loops of data/loader.py:22-33
 * (FileIO.load_data_set: one "user item weight" line per interaction, single-space separated)
 * and data/ui_graph.py:29-45 (ids in first-appearance order of the training file; test pairs kept
 * only when both ends occur in training).  Host only.
 * sizes5 = {n_users, n_items, n_train, n_test_kept, n_test_lines}; names are returned concatenated
 * (offsets has n+1 entries).  which: 0 = users, 1 = items.
 * ---------------------------------------------------------------------------------- */
typedef struct srh_dataset srh_dataset_t;
srh_status_t srh_dataset_load(srh_dataset_t** out, const char* train_path, const char* test_path /* or NULL */);
void srh_dataset_destroy(srh_dataset_t* ds);
srh_status_t srh_dataset_sizes(const srh_dataset_t* ds, int64_t* h_sizes5);
srh_status_t srh_dataset_copy_ids(const srh_dataset_t* ds, int32_t* h_train_u, int32_t* h_train_i,
                                  float* h_train_w, int32_t* h_test_u, int32_t* h_test_i);
int64_t srh_dataset_names_bytes(const srh_dataset_t* ds, int32_t which);
/* h_offsets != NULL: names back to back in h_buf (names_bytes bytes), name k = [h_offsets[k], h_offsets[k+1]); h_offsets
 * == NULL: every name followed by '\n' (names_bytes + count bytes; a name is a token of a line, so it holds no '\n') --
 * one bulk split on the caller's side instead of a slice per name. */
srh_status_t srh_dataset_copy_names(const srh_dataset_t* ds, int32_t which, char* h_buf, int64_t* h_offsets);

#ifdef __cplusplus
}
#endif
#endif /* SELFREC_HIP_H */
