// (a-3, a-4) CSR SpMM  y = A x  for LightGCN-family propagation, with the per-layer
// elementwise tail fused into the epilogue.  Replaces torch.sparse.mm at reference
// LightGCN.py:72, XSimGCL.py:88, SimGCL.py:85, SGL.py:104-108 and the perturb / stack /
// mean launches of XSimGCL.py:90-96.
//
// Bound: HBM / L2 gather bandwidth.  Algorithmic bytes per call:
//   nnz*8 (int32 col + fp32 val) + (n_rows+1)*4 + n_cols*d*4 (x read once) + n_rows*d*4 (y)
//
// Mapping for gfx950 (wave64) -- see the comment above spmm_rows_kernel:
//   * one embedding row is d fp32 = d/4 lanes x float4 (LPR lanes); a wave holds G = 64/LPR row-vectors side by
//     side (d=64: 16 lanes per row, 4 rows per wave); (col,val) pairs are loaded coalesced and broadcast inside a
//     16-lane DPP row; 8 gathers of 256-byte x rows are in flight per row-group before the first FMA.
//   * power-law rows: rows longer than split_len are cut into segments that publish partial sums with
//     write-through stores; the segment that arrives last (relaxed agent-scope ticket, cdna_hip_programming.md G16)
//     adds them in slot order -- run-to-run deterministic -- and applies the epilogue.  No second launch.
//   * XCD-aware issue order: workgroup b runs on XCD b % 8; the plan deals row classes (user / item rows of the
//     bipartite adjacency) and column classes (even / odd columns of long rows) to XCD groups so that each 4 MiB
//     L2 caches one part of x (measured: TCC hit rate 43 % -> 65 %, profiles/r01_a_pmc_*, r01_n_*).
//   * optional row / column activity marks skip whole rows (last forward layer: only the batch's rows are
//     needed) and zero columns (first backward layer: the incoming gradient is non-zero only on the batch's
//     rows); zero-valued entries never issue their gather.
// Earlier generations of this kernel (one wave per segment with ds_bpermute broadcasts, a persistent software-
// pipelined form, a two-launch split-row finish, lane-per-row thin tables) were measured and retired: their
// A/B numbers live in profiles/r01_* and DESIGN.md section 4.1, not in this library.
#include <algorithm>
#include <cstdlib>
#include <new>
#include <vector>

#include "common.h"

// Build-time switch (tools/spmm_lab/build_alt.sh builds the other setting for a file-level A/B of bench.py; no run-time knob
// ships):  SRH_Y_WT 1 (default): the epilogue's output rows (y, the layer mean, FANOUT copies) leave with write-through
// (`sc1`) stores.  A propagation launch writes 17.8 MB at the Yelp2018 shape that no workgroup of ITS launch reads again and
// that the next launch gathers from OTHER XCDs; left dirty in the writers' L2s it is written back at the kernel boundary
// (MI355X_MICROARCH.md price list, "boundary": + B / 6 TB/s).  Write-through moves those bytes under the gathers: dense
// launch 41.9 -> 41.1 us, step 0.2823 -> 0.2792 ms in a same-box A/B (profiles/r04_b_write_through_ab.txt).
#ifndef SRH_Y_WT
#define SRH_Y_WT 1
#endif

namespace {

using namespace srh;
constexpr bool kYWT = SRH_Y_WT != 0;

struct Seg {
  int32_t row, start, end, slot;  // slot < 0: final result of `row`; else partial[slot]
};
struct Heavy {
  int32_t row, first_slot, n_slots, pad;
};
// slot -> index into the Heavy array (one int per partial slot)

struct DevEpilogue {
  int32_t flags;
  float eps;
  const float* noise;
  uint32_t seed_lo, seed_hi, off_lo, off_hi;
  const int64_t* rng_step;
  uint64_t rng_stride;
  float alpha;
  int32_t n_prev, n_add;
  const float* prev[SRH_MAX_PREV];
  float mean_rcp;
  float* mean_out;
  const float* add[SRH_MAX_ADD];
  float add_scale[SRH_MAX_ADD];
  const int32_t* row_mark;
  const int32_t* col_mark;
  const int64_t* mark_stamp;
  const int32_t* add_mark;       // AXPY: addends flagged in add_sparse are zero outside marked rows
  int32_t add_sparse;
  // FANOUT: further perturbed copies of the same product
  int32_t n_extra, main_clean;
  float* extra_out[SRH_MAX_EXTRA];
  const float* extra_noise[SRH_MAX_EXTRA];
  uint32_t extra_off_lo[SRH_MAX_EXTRA], extra_off_hi[SRH_MAX_EXTRA];
  // column-sharded tables (thin kernel): y holds columns [noise_col0, noise_col0 + d) of rows that are
  // noise_d_full wide -- the perturbation's unit vector is normalised over the WHOLE row
  int32_t noise_d_full, noise_col0;
  // rows padded with zero columns (any embedding.size): only the first noise_d_valid columns of the WHOLE row carry
  // noise -- the unit vector is normalised over those (the padding's y is exactly 0, so sign(y) keeps it 0)
  int32_t noise_d_valid;
  // row scaling (value-free products: include/selfrec_hip.h)
  const float* row_scale;
  int32_t scale_flags, prev_unscale, add_rowscale;
  unsigned long long* stamps;    // srh_spmm_f32_probe: {begin, end, XCD} per wave (PROBE instantiations only)
  // ADAM: the finished row is the gradient of adam_p's row -- updated here, the gradient is never stored
  float4 *adam_p, *adam_m, *adam_v;
  const float* adam_coef;        // {lr / bias_correction1, sqrt(bias_correction2)} of this step (batch_fetch_body)
  float adam_b1, adam_b2, adam_eps;
  int32_t adam_n_clear;
  const int32_t* adam_clear_mark;
  float4* adam_clear[SRH_MAX_ADAM_CLEAR];
  int64_t* adam_cursor;
};

// Counter-based noise: every element's uniform is a pure function of (seed, counter, element),
// so there is no generator state in memory and a captured graph replays with fresh noise by
// bumping one device-side counter.  The mixer is the 2-multiply "lowbias32" integer hash (full
// avalanche, bias < 0.2 bits): the perturbation only needs decorrelated U[0,1) draws, and
// profiling showed the row epilogue -- not memory -- bounding the kernel, where Philox4x32-10
// costs ~100 VALU ops per float4 against ~35 here.
__device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint4 counter_rng4(uint64_t ctr, uint32_t sub, uint32_t seed_lo, uint32_t seed_hi) {
  const uint32_t key = lowbias32((uint32_t)ctr ^ seed_lo) + lowbias32((uint32_t)(ctr >> 32) ^ seed_hi);
  const uint32_t base = key + sub * 0x9E3779B1U;
  return make_uint4(lowbias32(base), lowbias32(base + 0x85EBCA6BU), lowbias32(base + 0xC2B2AE35U),
                    lowbias32(base + 0x27D4EB2FU));
}
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

// Epilogue on one full row held as float4 per lane of an LPR-lane group.  Executed by all
// groups of the wave with identical data (they all hold the reduced row); `store` selects
// the group that writes.
// Partials another XCD published with a write-through store are read with agent-scope (sc1) loads: coherence per
// access.  An acquire FENCE instead (buffer_inv sc1) drops the whole XCD L2 -- thousands of them per launch were
// costing the x rows their hit rate.
// The partial sums of a split row, added in slot order by the segment that arrived last: partials t0, t0 + step, ...
// (< hn) of this lane's 16 bytes.  Eight 16-byte agent-scope loads are in flight before the first add (one
// global_load_dwordx4 sc1 each; 16-byte sc1 accesses are observed untorn on gfx950, MI355X_MICROARCH.md): the plain loop
// -- four dword loads, wait, add, per partial -- put up to 30 dependent round trips at the end of the heaviest rows, which
// is the critical path of the batch-masked launches (profiles/r02_i_wave_timeline_*).  Same order of additions.
__device__ __forceinline__ float4 sum_partials_agent(const float4* base, int t0, int step, int hn, int stride4) {
  typedef float fx4 __attribute__((ext_vector_type(4)));
  float4 sum = f4_zero();
  int t = t0;
  for (; t + 7 * step < hn; t += 8 * step) {
    fx4 p[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
      asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(p[k]) : "v"(base + (size_t)(t + k * step) * stride4) : "memory");
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7])::"memory");
#pragma unroll
    for (int k = 0; k < 8; ++k) sum = f4_add(sum, make_float4(p[k].x, p[k].y, p[k].z, p[k].w));
  }
  for (; t < hn; t += step) {
    fx4 q;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(q) : "v"(base + (size_t)t * stride4) : "memory");
    sum = f4_add(sum, make_float4(q.x, q.y, q.z, q.w));
  }
  return sum;
}

// columns >= dv of float4 number q of a row carry no noise (zero-padded rows)
__device__ __forceinline__ float4 mask_valid(float4 z, int q, int dv) {
  const int c = 4 * q;
  return make_float4(c < dv ? z.x : 0.f, c + 1 < dv ? z.y : 0.f, c + 2 < dv ? z.z : 0.f, c + 3 < dv ? z.w : 0.f);
}

// y + sign(y) * normalize(noise_row) * eps  (XSimGCL.py:90-91); noise injected or from the counter RNG
template <int LPR>
__device__ __forceinline__ float4 perturb_row(float4 y, int row, int sub, size_t at, const float* noise,
                                              uint32_t off_lo, uint32_t off_hi, const DevEpilogue& ep) {
  float4 nu;
  float ss;
  if (ep.noise_d_full != 4 * LPR) {
    // y is a column slice of noise_d_full-wide rows: the unit vector is normalised over the WHOLE row, whose
    // other columns are regenerated (counter RNG) or read (injected noise) by the group's lanes in turn
    const int nq = ep.noise_d_full >> 2, own = (ep.noise_col0 >> 2) + sub;
    uint64_t ctr = (((uint64_t)off_hi << 32) | off_lo) + (uint64_t)row;
    if (!noise && ep.rng_step) ctr += (uint64_t)(*ep.rng_step) * ep.rng_stride;
    const float4* nr = reinterpret_cast<const float4*>(noise) + (size_t)row * nq;
    const bool padded = ep.noise_d_valid < ep.noise_d_full;
    auto draw = [&](int q) {
      float4 z;
      if (noise) {
        z = nr[q];
      } else {
        const uint4 r = counter_rng4(ctr, (uint32_t)q, ep.seed_lo, ep.seed_hi);
        z = make_float4(u01(r.x), u01(r.y), u01(r.z), u01(r.w));
      }
      return padded ? mask_valid(z, q, ep.noise_d_valid) : z;
    };
    ss = 0.f;
    for (int q = sub; q < nq; q += LPR) { const float4 z = draw(q); ss += f4_dot(z, z); }
    ss = group_sum<LPR>(ss);
    nu = draw(own);
  } else {
    if (noise) {
      nu = reinterpret_cast<const float4*>(noise)[at];
    } else {
      uint64_t ctr = (((uint64_t)off_hi << 32) | off_lo) + (uint64_t)row;
      if (ep.rng_step) ctr += (uint64_t)(*ep.rng_step) * ep.rng_stride;
      uint4 r = counter_rng4(ctr, (uint32_t)sub, ep.seed_lo, ep.seed_hi);
      nu = make_float4(u01(r.x), u01(r.y), u01(r.z), u01(r.w));
    }
    if (ep.noise_d_valid < 4 * LPR) nu = mask_valid(nu, sub, ep.noise_d_valid);      // (uniform: zero-padded rows)
    ss = group_sum<LPR>(f4_dot(nu, nu));
  }
  // F.normalize: v / max(||v||, 1e-12); one reciprocal instead of four divisions (<= 1 ulp apart)
  const float scale = ep.eps / fmaxf(sqrtf(ss), 1e-12f);
  y.x += sgnf(y.x) * (nu.x * scale);
  y.y += sgnf(y.y) * (nu.y * scale);
  y.z += sgnf(y.z) * (nu.z * scale);
  y.w += sgnf(y.w) * (nu.w * scale);
  return y;
}

// r: the row's scale factor (1 when the launch has none) -- loaded by the caller BEFORE its gathers: these waves run on
// their chain of dependent round trips, and a load issued here would add one to every row
template <int LPR, bool ADAM = false>
__device__ __forceinline__ void row_epilogue(float4 y, int row, int sub, bool store, float4* __restrict__ Y,
                                             const DevEpilogue& ep, const float r = 1.0f) {
  // (`store` is uniform over the row-group: a group that does not write -- groups 1.. of a cooperative task, the
  // row-groups of dead rows on a row-masked launch -- leaves before the addend / layer-mean loads and the noise hash)
  if (!store) return;
  const size_t at = (size_t)row * LPR + sub;
  float4 mm, vv, pp;
  if constexpr (ADAM) {     // (first: in flight together with the addends below, not a round trip after them)
    mm = ep.adam_m[at]; vv = ep.adam_v[at]; pp = ep.adam_p[at];
  }
  if (ep.scale_flags & SRH_SCALE_IN) y = f4_scale(y, r);
  if (ep.flags & SRH_EPI_AXPY) {
    y = f4_scale(y, ep.alpha);
    // addends that are non-zero only on this step's batch rows are not even read elsewhere
    const bool marked = !ep.add_mark || ep.add_mark[row] == (int)(*ep.mark_stamp);
    for (int t = 0; t < ep.n_add; ++t) {
      if (((ep.add_sparse >> t) & 1) && !marked) continue;
      float4 a = reinterpret_cast<const float4*>(ep.add[t])[at];
      const float sc = ((ep.add_rowscale >> t) & 1) ? ep.add_scale[t] * r : ep.add_scale[t];
      y = f4_fma(sc, a, y);
    }
  }
  if constexpr (ADAM) {
    // (its own instantiation: the DevEpilogue fields below would otherwise sit in the SGPRs of every product)
    // y = d loss / d param[row]: adam_kernel's update (optim.hip), operation for operation, on this lane's 16 bytes of
    // the row; the 71 MB a separate pass spends writing and re-reading the gradient are never moved
    const float step_size = ep.adam_coef[0], bc2_sqrt = ep.adam_coef[1];
    const float b1 = ep.adam_b1, b2 = ep.adam_b2, omb1 = 1.0f - b1, omb2 = 1.0f - b2, aeps = ep.adam_eps;
#define SRH_ADAM_LANE(c) adam_element(mm.c, vv.c, pp.c, y.c, b1, omb1, b2, omb2, step_size, bc2_sqrt, aeps);
    SRH_ADAM_LANE(x) SRH_ADAM_LANE(y) SRH_ADAM_LANE(z) SRH_ADAM_LANE(w)
#undef SRH_ADAM_LANE
    st_f4<kYWT>(ep.adam_m + at, mm); st_f4<kYWT>(ep.adam_v + at, vv); st_f4<kYWT>(ep.adam_p + at, pp);
    // this step's rows of the batch-sparse gradient buffers: read (as addends) by this row's task alone, above
    if (ep.adam_n_clear > 0 && ep.adam_clear_mark[row] == (int)(*ep.mark_stamp)) {
      for (int k = 0; k < ep.adam_n_clear; ++k) ep.adam_clear[k][at] = f4_zero();
    }
    return;
  }
  const bool out_scaled = (ep.scale_flags & SRH_SCALE_OUT) != 0;
  if (ep.flags & SRH_EPI_PERTURB) {
    const float4 raw = y;
    if (!ep.main_clean) y = perturb_row<LPR>(raw, row, sub, at, ep.noise, ep.off_lo, ep.off_hi, ep);
    for (int k = 0; k < ep.n_extra; ++k) {
      const float4 yk = perturb_row<LPR>(raw, row, sub, at, ep.extra_noise[k], ep.extra_off_lo[k], ep.extra_off_hi[k], ep);
      if (store) st_f4<kYWT>(reinterpret_cast<float4*>(ep.extra_out[k]) + at, out_scaled ? f4_scale(yk, r) : yk);
    }
  }
  if (store) st_f4<kYWT>(Y + at, out_scaled ? f4_scale(y, r) : y);
  if (ep.flags & SRH_EPI_MEAN) {
    const float rinv = (ep.prev_unscale && r > 0.f) ? 1.0f / r : 0.f;
    float4 m = y;
    // (earlier layers first, this layer last: the reference's stack order -- and, without unscaled terms, the
    // summation order of the previous form of this loop, bit for bit)
    if (ep.n_prev > 0) {
      m = reinterpret_cast<const float4*>(ep.prev[0])[at];
      if (ep.prev_unscale & 1) m = f4_scale(m, rinv);
      for (int t = 1; t < ep.n_prev; ++t) {
        float4 pt = reinterpret_cast<const float4*>(ep.prev[t])[at];
        if ((ep.prev_unscale >> t) & 1) pt = f4_scale(pt, rinv);
        m = f4_add(m, pt);
      }
      m = f4_add(m, y);
    }
    if (store) {
      float4 o = f4_scale(m, ep.mean_rcp);
      st_f4<kYWT>(reinterpret_cast<float4*>(ep.mean_out) + at, o);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// spmm_rows_kernel: one task per wave.
//   * (col,val) are broadcast inside each 16-lane DPP row with v_mov_dpp row_newbcast (one VALU op, no LDS
//     traffic): the lanes of a DPP row hold 16 consecutive entries and round t uses lane t;
//   * every row-group gets its OWN short row (<= 64 non-zeros; 4 rows per wave at d=64), so the epilogue --
//     noise, normalisation, mean, stores -- runs once per wave for G rows and needs no cross-group reduction;
//     rows are issued longest-first, so the G rows of a wave have (nearly) the same length;
//   * one wave per long row / split segment ("coop" tasks): each row-group takes a block of 16 entries of a
//     16*G-entry chunk and the groups are summed at the end.
// ---------------------------------------------------------------------------------------------
typedef float floatx4_t __attribute__((ext_vector_type(4)));

struct Task {
  int32_t kind, first, count, pad;   // kind 0: coop on segs[first]; kind 1: rows segs[first .. first+count)
};
constexpr int kShortRow = 64;

template <int K>
__device__ __forceinline__ int row_bcast_i(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x150 + K, 0xf, 0xf, false); }
template <int K>
__device__ __forceinline__ float row_bcast_f(float x) { return __int_as_float(row_bcast_i<K>(__float_as_int(x))); }

// x row `c`, this lane's 16 bytes: uniform 64-bit base + 32-bit byte offset (one VALU op instead of a
// sign-extend, a 64-bit shift and a 64-bit add per gather; srh_spmm_f32 checks the table is < 4 GiB)
template <int LPR>
__device__ __forceinline__ float4 ld_x(const float4* __restrict__ X, int c, int sub) {
  const unsigned off = (unsigned)c * (unsigned)(LPR * 16) + (unsigned)(sub * 16);
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(X) + off);
}

// 16-byte write-through store (sc1): the partial goes straight to memory and is not left dirty in this
// XCD's L2, so publishing it needs no L2 write-back fence (cdna_hip_programming.md G16, form R1)
__device__ __forceinline__ void store_f4_sc1(float4* p, float4 v) {
  floatx4_t x = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");   // (nop: see st_f4, common.h)
}

// ---- the gather loop of spmm_rows_kernel, in inline assembly -------------------------------------------
// What the compiler makes of the C++ form (profiles/r02_a_spmm_lab.txt, variant 0): ~12 VALU + 4 SALU
// instructions per gathered x row -- a zero-init mov for each DPP broadcast's `old` operand, a v_lshl_or per
// address, four zero-fill movs + saveexec / branch / restore around every predicated load, and one even-aligned
// VGPR PAIR per broadcast value for v_pk_fma's 64-bit operand (8 pairs: the kernel sat at 70 VGPRs).  Here, per
// entry: address = v_or_b32_dpp(col * row bytes, lane offset) -- DPP broadcast and address arithmetic in ONE
// instruction --, predicate = v_cmpx on the offset's sign bit (padding / dropped edges / dead columns carry it:
// they issue no gather, their destination keeps an older, FINITE x row that is then multiplied by 0), value =
// one v_mov_b32_dpp issued after the loads (VALU under the memory latency), two v_pk_fma_f32 with two entries'
// values sharing one register pair (op_sel).  5 VALU + 1 SALU per entry, 62 VGPRs: 8 waves per SIMD.
// Inline asm is invisible to the compiler's hazard recogniser and waitcnt insertion, hence the explicit s_nop
// (VALU write -> DPP read: 2 wait states; VALU-written exec -> DPP: 5) and the explicit vmcnt counts; a load the
// compiler issues in between only makes those waits conservative (returns are in order).
typedef float floatx2_t __attribute__((ext_vector_type(2)));
struct Acc2 { floatx2_t lo, hi; };       // columns 0-1 / 2-3 of this lane's float4 of the output row

#define SRH_DPP_OR(T)                                                                                          \
  asm volatile("v_or_b32_dpp %0, %1, %2 row_newbcast:" #T " row_mask:0xf bank_mask:0xf" : "=v"(off[T & 7]) : "v"(cs), "v"(sub16))
#define SRH_DPP_MOV(T)                                                                                         \
  asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:" #T " row_mask:0xf bank_mask:0xf" : "=v"(vv[T & 7]) : "v"(v))
// acc += value * x for one gathered row, as soon as it has landed (N younger loads may still be in flight)
#define SRH_FMA(N, SEL, VP, XR)                                                                                 \
  asm volatile("s_waitcnt vmcnt(" #N ")\n\t"                                                                    \
               "v_pk_fma_f32 %[lo], %[vp], %[xlo], %[lo] op_sel:[" #SEL ",0,0] op_sel_hi:[" #SEL ",1,1]\n\t"      \
               "v_pk_fma_f32 %[hi], %[vp], %[xhi], %[hi] op_sel:[" #SEL ",0,0] op_sel_hi:[" #SEL ",1,1]"          \
               : [lo] "+v"(acc.lo), [hi] "+v"(acc.hi)                                                           \
               : [vp] "v"(VP), [xlo] "v"(__builtin_shufflevector(XR, XR, 0, 1)), [xhi] "v"(__builtin_shufflevector(XR, XR, 2, 3)))

// Eight entries of the DPP row (lanes 0-7 or 8-15), of which only the LAST `nr` carry an entry (nr = 8: a whole half).
// Every round a chunk issues beyond the ones it needs is a vector-memory instruction paid for nothing: with whole
// eight-round halves only, the plan of the Yelp2018-shape graph issued 762 k gathers per launch for 630 k x four rows
// (profiles/r02_h_*).  The kernel therefore packs a tail's entries into the last nr lanes of the half (lanes 8 - nr .. 7,
// see lane_slot in spmm_rows_kernel) and the load block jumps over the first 8 - nr loads.  Loads are issued in lane
// order, multiply-adds follow in the same order with vmcnt(7 .. 0) -- a skipped load is an OLDER load that was never
// issued, so the counts of the issued ones stand; its destination keeps an older finite x row and its value is 0.
// Entries are summed in ascending order (the whole-half form this replaced gave the same sums bit for bit).
#define SRH_PL(K) ".Lsrh_t" #K "_%=:\n\tv_cmpx_le_i32_e32 0, %[o" #K "]\n\tglobal_load_dwordx4 %[x" #K "], %[o" #K "], %[b]\n\ts_mov_b64 exec, %[sv]\n\t"
#define SRH_GE(N, K) "s_cmp_ge_i32 %[nr], " #N "\n\ts_cbranch_scc1 .Lsrh_t" #K "_%=\n\t"
__device__ __forceinline__ void pred_load8_tail(int nr, floatx4_t& x0, floatx4_t& x1, floatx4_t& x2, floatx4_t& x3,
                                                floatx4_t& x4, floatx4_t& x5, floatx4_t& x6, floatx4_t& x7,
                                                const unsigned (&off)[8], const void* X) {
  unsigned long long save;
  asm volatile("s_mov_b64 %[sv], exec\n\t"
               SRH_GE(8, 0) SRH_GE(7, 1) SRH_GE(6, 2) SRH_GE(5, 3) SRH_GE(4, 4) SRH_GE(3, 5) SRH_GE(2, 6)
               "s_branch .Lsrh_t7_%=\n\t"
               SRH_PL(0) SRH_PL(1) SRH_PL(2) SRH_PL(3) SRH_PL(4) SRH_PL(5) SRH_PL(6) SRH_PL(7)
               "s_nop 4"
               : [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2), [x3] "+v"(x3), [x4] "+v"(x4), [x5] "+v"(x5), [x6] "+v"(x6),
                 [x7] "+v"(x7), [sv] "=&s"(save)
               : [o0] "v"(off[0]), [o1] "v"(off[1]), [o2] "v"(off[2]), [o3] "v"(off[3]), [o4] "v"(off[4]), [o5] "v"(off[5]),
                 [o6] "v"(off[6]), [o7] "v"(off[7]), [b] "s"(X), [nr] "s"(nr)
               : "memory", "vcc", "scc");
}
#undef SRH_PL
#undef SRH_GE

// nr (1 .. 8): rounds of THIS half (lanes 0-7 or 8-15 of the DPP row) that carry an entry somewhere in the wave: the last nr
struct NoIssue { __device__ __forceinline__ void operator()() const {} };
template <bool HI, bool PF = false, class F = NoIssue>
__device__ __forceinline__ void gather8_tail(int nr, unsigned cs, float v, unsigned sub16, const void* X, floatx4_t (&xx)[8],
                                             Acc2& acc, F issue_next = F()) {
  unsigned off[8];
  float vv[8];
  asm volatile("s_nop 1" : "+v"(cs), "+v"(v));
  if (!HI) {
    SRH_DPP_OR(0); SRH_DPP_OR(1); SRH_DPP_OR(2); SRH_DPP_OR(3); SRH_DPP_OR(4); SRH_DPP_OR(5); SRH_DPP_OR(6); SRH_DPP_OR(7);
  } else {
    SRH_DPP_OR(8); SRH_DPP_OR(9); SRH_DPP_OR(10); SRH_DPP_OR(11); SRH_DPP_OR(12); SRH_DPP_OR(13); SRH_DPP_OR(14); SRH_DPP_OR(15);
  }
  pred_load8_tail(nr, xx[0], xx[1], xx[2], xx[3], xx[4], xx[5], xx[6], xx[7], off, X);
  if (PF) {
    issue_next();                       // the next chunk's raw (col [, val]): YOUNGER than this batch's gathers
    asm volatile("s_nop 4" ::: "memory");
  }
  if (!HI) {
    SRH_DPP_MOV(0); SRH_DPP_MOV(1); SRH_DPP_MOV(2); SRH_DPP_MOV(3); SRH_DPP_MOV(4); SRH_DPP_MOV(5); SRH_DPP_MOV(6); SRH_DPP_MOV(7);
  } else {
    SRH_DPP_MOV(8); SRH_DPP_MOV(9); SRH_DPP_MOV(10); SRH_DPP_MOV(11); SRH_DPP_MOV(12); SRH_DPP_MOV(13); SRH_DPP_MOV(14); SRH_DPP_MOV(15);
  }
  const floatx2_t p0 = {vv[0], vv[1]}, p1 = {vv[2], vv[3]}, p2 = {vv[4], vv[5]}, p3 = {vv[6], vv[7]};
  if (PF) {
    SRH_FMA(8, 0, p0, xx[0]); SRH_FMA(7, 1, p0, xx[1]); SRH_FMA(6, 0, p1, xx[2]); SRH_FMA(5, 1, p1, xx[3]);
    SRH_FMA(4, 0, p2, xx[4]); SRH_FMA(3, 1, p2, xx[5]); SRH_FMA(2, 0, p3, xx[6]); SRH_FMA(1, 1, p3, xx[7]);
  } else {
    SRH_FMA(7, 0, p0, xx[0]); SRH_FMA(6, 1, p0, xx[1]); SRH_FMA(5, 0, p1, xx[2]); SRH_FMA(4, 1, p1, xx[3]);
    SRH_FMA(3, 0, p2, xx[4]); SRH_FMA(2, 1, p2, xx[5]); SRH_FMA(1, 0, p3, xx[6]); SRH_FMA(0, 1, p3, xx[7]);
  }
}

// The same eight entries in plain C++, for COLUMN-MASKED launches (first backward layer: more than half of the
// entries are dead): the compiler's version branches over a gather whose whole wave is dead, where the asm form
// still issues the exec = 0 load -- measured 34.2 against 38.1 us at the Yelp2018 shape (profiles/r02_a_spmm_lab.txt).
template <int LPR, int T0>
__device__ __forceinline__ void gather8_branchy(int c, float v, const float4* __restrict__ X, int sub, float4& acc) {
  int cc[8];
  float vv[8];
  float4 xx[8];
#define SRH_BC(T) cc[T] = row_bcast_i<T0 + T>(c); vv[T] = row_bcast_f<T0 + T>(v);
  SRH_BC(0) SRH_BC(1) SRH_BC(2) SRH_BC(3) SRH_BC(4) SRH_BC(5) SRH_BC(6) SRH_BC(7)
#undef SRH_BC
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    xx[t] = f4_zero();
    if (vv[t] != 0.f) xx[t] = ld_x<LPR>(X, cc[t], sub);
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) acc = f4_fma(vv[t], xx[t], acc);
}

// Column-masked launches with the live entries COMPACTED first.  Three quarters of the entries of that launch are dead
// (24 % of the non-zeros sit in the batch's columns at the Yelp2018 shape), and gather8_branchy still spends ~16
// instructions on each of them -- two DPP broadcasts, a compare / saveexec / branch, four multiply-adds by zero, four
// zero-fill movs: with NO live column the launch takes 22 of its 28 us, and neither shorter latency chains nor
// coalesced per-entry marks moved that (profiles/r02_e_masked_launches.txt): it is instruction issue.  Here the 16
// entries a DPP row holds are first permuted so that the live ones occupy lanes 0 .. n-1 (a ballot, two popcounts and
// one ds_permute each for column and value), and the broadcast rounds stop at the fullest row's count (uniform): ~4-6
// rounds instead of 16.  Same entries, same order within a row => the same sums, bit for bit.
template <int LPR, int T0>
__device__ __forceinline__ void gather4_live(int c, float v, const float4* __restrict__ X, int sub, float4& acc) {
  int cc[4];
  float vv[4];
  float4 xx[4];
#define SRH_BC(T) cc[T] = row_bcast_i<T0 + T>(c); vv[T] = row_bcast_f<T0 + T>(v);
  SRH_BC(0) SRH_BC(1) SRH_BC(2) SRH_BC(3)
#undef SRH_BC
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    xx[t] = f4_zero();
    if (vv[t] != 0.f) xx[t] = ld_x<LPR>(X, cc[t], sub);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) acc = f4_fma(vv[t], xx[t], acc);
}

template <int LPR>
__device__ __forceinline__ void gather16_compact(int c, float v, const float4* __restrict__ X, int sub, float4& acc) {
  const int lane = threadIdx.x & 63, e16 = lane & 15, row0 = lane & 48;
  const bool alive = v != 0.f;
  const unsigned long long bal = __ballot(alive);
  if (bal == 0ull) return;
  const unsigned gm = (unsigned)(bal >> row0) & 0xffffu;           // this DPP row's live lanes
  const int n_g = __popc(gm), before = __popc(gm & ((1u << e16) - 1u));
  // a permutation of the row: live lanes to the front in order, dead lanes behind them
  const int dst = row0 + (alive ? before : n_g + (e16 - before));
  c = __builtin_amdgcn_ds_permute(dst << 2, c);
  v = __int_as_float(__builtin_amdgcn_ds_permute(dst << 2, __float_as_int(v)));
  const int nmax = max(max(__popcll(bal & 0xffffull), __popcll((bal >> 16) & 0xffffull)),
                       max(__popcll((bal >> 32) & 0xffffull), __popcll(bal >> 48)));
  gather4_live<LPR, 0>(c, v, X, sub, acc);
  if (nmax > 4) gather4_live<LPR, 4>(c, v, X, sub, acc);
  if (nmax > 8) gather4_live<LPR, 8>(c, v, X, sub, acc);
  if (nmax > 12) gather4_live<LPR, 12>(c, v, X, sub, acc);
}

// One record per wave, read with scalar loads (the task -> segment -> (col, val) chain of dependent VECTOR round
// trips loses its first two links): kind 0 = the wave's cooperative segment (row[0], start[0], end[0], slot),
// kind 1 = `count` <= G short rows, one per row-group.
struct alignas(64) Task64 {
  int32_t kind, count, slot, pad;
  int32_t row[4], start[4], end[4];
};

// Split rows are completed inside this launch by whichever of their segments arrives last (write-through
// partials -> vmcnt(0) -> relaxed agent-scope ticket; the last arriver reads the partials with agent-scope
// loads, adds them in slot order -- bitwise reproducible -- runs the epilogue and re-arms the ticket).
// COLMASK: the launch carries column activity marks (its own instantiation: the unmasked kernel then has no mark
// code and always prefetches; a run-time switch between the two cost 2.5 us per launch).
// A chunk's tail issues only the rounds it needs (gather8_tail), and a cooperative task deals its entries to the row-
// groups round-robin (entry k of a chunk -> group k % G, round k / G), so that a tail of R entries needs ceil(R / G) rounds
// with every group busy rather than up to 16 rounds with one.
// PROBE (srh_spmm_f32_probe): every wave also leaves {begin, end} on the chip-wide 100 MHz clock and the XCD it ran on
// in stamps[3 * wave ..] -- what the engine's start-up calibration of the plan's XCD shares reads (engine.py).
// (Round 4, measured and rejected: sixteen gathers in flight per row-group for this flavour -- 128 VGPRs, 4 waves per SIMD:
// row-masked launch 19.0 -> 28.4 us, at 3 waves per SIMD 33.6: the live waves are bound by how many of them the chip holds,
// not by the length of a wave's chain of gather batches; profiles/r04_b_write_through_ab.txt.)
// LATEPF (round 3): the next chunk's (col, val) are issued AFTER the current chunk's gathers instead of before them.  An
// A/B of the whole library settled where it belongs (profiles/r03_b_late_prefetch_ab.txt): the dense launches lose 0.4-0.6
// us with it (their waves overlap enough for the early prefetch to be free), the ROW-MASKED launch -- a few thousand live
// cooperative waves on a chain of dependent gather batches -- gains 1.5-1.8 us: instantiated for that flavour only.
// FLOOR (srh_spmm_gather_bound): the launch's own lower bound, measured -- the SAME task list on the SAME workgroup ->
// XCD placement, the same (col) stream and the same eight-in-flight gathers of x rows with their multiply-adds, and
// nothing after them: no values (pattern), no cross-group reduction, no split-row hand-off, no epilogue, no y.  A strict
// subset of the product's work on the product's schedule, so it cannot come out slower than the product it bounds
// (round 4's stand-alone gather probe walked the index array in its own order and did: 48 us against 40).
// ADAM (srh_spmm_epilogue_t, SRH_EPI_ADAM): the row epilogue ends in the optimiser's update of the row instead of a store of
// y, and wave 0 advances the step cursor.
template <int LPR, bool COLMASK, bool PROBE = false, bool LATEPF = false, bool FLOOR = false, bool ADAM = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void spmm_rows_kernel(const Task64* __restrict__ tasks, int n_tasks,
                                                        const int32_t* __restrict__ indices,
                                                        const float* __restrict__ vals,
                                                        const float4* __restrict__ X, float4* __restrict__ Y,
                                                        float4* __restrict__ partial,
                                                        const Heavy* __restrict__ heavy,
                                                        const int32_t* __restrict__ slot_owner,
                                                        int32_t* __restrict__ tickets, DevEpilogue ep,
                                                        const int n_fetch, const srh_batch_fetch_args_t rider) {
  constexpr int G = 64 / LPR;          // row-groups per wave
  constexpr int CH = 16 * G;           // entries per coop chunk
  // srh_spmm_f32_with_fetch: the first n_fetch workgroups (a multiple of 8: every task keeps its XCD) stage the step's
  // batch instead -- nothing in this product reads what they write
  if ((int)blockIdx.x < n_fetch) {
    srh::batch_fetch_body(rider, (int)blockIdx.x, n_fetch);
    return;
  }
  const int wave = __builtin_amdgcn_readfirstlane((int)(((blockIdx.x - n_fetch) * 256u + threadIdx.x) >> 6));
  if (wave >= n_tasks) return;
  const int lane = threadIdx.x & 63;
  // (ADAM: the last launch of the step moves the cursor; everything here reads the step from batch_fetch's copy)
  if constexpr (ADAM) {
    if (ep.adam_cursor && wave == 0 && lane == 0) { ep.adam_cursor[0] += 1; ep.adam_cursor[1] += 1; }
  }
  unsigned long long t_begin = 0;
  unsigned xcc = 0;
  if constexpr (PROBE) {
    t_begin = wall_clock64();
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  }
  auto leave = [&]() {
    if constexpr (PROBE) if (lane == 0) {
      ep.stamps[3 * (size_t)wave] = t_begin;
      ep.stamps[3 * (size_t)wave + 1] = wall_clock64();
      ep.stamps[3 * (size_t)wave + 2] = xcc & 0xfu;
    }
  };
  const int g = lane / LPR, sub = lane % LPR, e16 = lane & 15;
  const unsigned sub16 = (unsigned)sub * 16u;
  const int stamp = ep.mark_stamp ? (int)(*ep.mark_stamp) : 0;
  const floatx4_t zero = {0.f, 0.f, 0.f, 0.f};
  Acc2 acc = {{0.f, 0.f}, {0.f, 0.f}};
  floatx4_t xx[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) xx[t] = zero;

  const Task64* tp = tasks + wave;       // uniform address: s_load
  const int kind = tp->kind, count = tp->count, slot = tp->slot;
  int row = tp->row[0], s = tp->start[0], e = tp->end[0];
  if (kind == 1 && G > 1) {
    const int r1 = tp->row[1], s1 = tp->start[1], e1 = tp->end[1];
    if (g == 1) { row = r1; s = s1; e = e1; }
    if (G > 2) {
      const int r2 = tp->row[2], s2 = tp->start[2], e2 = tp->end[2], r3 = tp->row[3], s3 = tp->start[3], e3 = tp->end[3];
      if (g == 2) { row = r2; s = s2; e = e2; }
      if (g == 3) { row = r3; s = s3; e = e3; }
    }
  }
  // (col, val) of one entry as the gather wants them: the column pre-multiplied by the row bytes, the sign bit on
  // entries that must not gather (padding, zero values = SGL's dropped edges, columns dead this step)
  auto fetch = [&](int j, int end, unsigned& cs, float& v) {
    int c = 0;
    v = 0.f;
    if (j < end) { c = indices[j]; v = vals ? vals[j] : 1.0f; }      // vals == NULL: pattern matrix
    if (COLMASK) {
      if (v != 0.f && ep.col_mark[c] != stamp) v = 0.f;
      cs = (unsigned)c;                                   // (gather8_branchy takes the plain column)
    } else {
      cs = (v == 0.f) ? 0x80000000u : (unsigned)c * (unsigned)(LPR * 16);
    }
  };
  // the next chunk's (col, val) are in flight under the current gathers -- except on column-masked launches, where
  // the mark gather hangs off the column load and prefetching two dependent loads was measured to lose
  constexpr bool prefetch = !COLMASK;
  unsigned cs, csn = 0x80000000u;
  float v, vn = 0.f;
  float4 accm = f4_zero();                                // COLMASK accumulator
  constexpr bool TAIL = !COLMASK;                         // tails issue only the rounds they need
  constexpr bool INTER = TAIL;                            // cooperative entries dealt round-robin to the groups
  // nr: broadcast rounds the chunk needs (>= 1; 16 and more = all)
  auto chunk = [&](int nr) {
    if (COLMASK) {
      gather16_compact<LPR>((int)cs, v, X, sub, accm);
    } else {
      // (two asm bodies per call site, not three: with a plain gather8 for whole first halves next to them the register
      // allocator no longer finds eight aligned quads under the 64-VGPR cap and spills 160 bytes into the loop)
      gather8_tail<false>(min(nr, 8), cs, v, sub16, X, xx, acc);
      if (nr > 8) gather8_tail<true>(min(nr, 16) - 8, cs, v, sub16, X, xx, acc);
    }
  };
  // The entry slot (= broadcast round, 0 .. 15) this lane's DPP-row position holds in a chunk that needs nr rounds, or -1.
  // A whole chunk: slot = lane.  A tail packs its nr slots into the LAST lanes of the half they end in (gather8_tail
  // skips the leading loads): nr <= 8 -> lanes 8 - nr .. 7; 8 < nr < 16 -> lanes 0 .. 7 and 24 - nr .. 15.
  auto lane_slot = [&](int nr) {
    if (!TAIL) return e16;
    const int shift = (e16 < 8) ? max(0, 8 - nr) : ((nr > 8) ? max(0, 16 - nr) : 16);
    const int slot = e16 - shift;
    return (slot >= (e16 & 8)) ? slot : -1;
  };
  // LATE: the next chunk's (col, val) are issued after the current chunk's last gathers (second half, always: a clamped
  // valid position on a task's last chunk), so that the gathers' returns do not queue behind that miss
  constexpr bool LATE = TAIL && LATEPF;
  int c_raw = 0;
  float v_raw = 0.f;
  auto chunk_late = [&](int nr, int jn, int end) {
    // The prefetch loads are issued INSIDE inline asm: the multiply-adds of the PF form wait with vmcnt(8 .. 1),
    // which is only right while these loads sit between the gathers and the multiply-adds -- a plain C++ load of a
    // `const __restrict__` array may legally be moved past an asm block by the compiler, and vmcnt(8) would then stop
    // waiting for the oldest gather.  asm volatile statements keep their order; the pair is first read after an explicit
    // vmcnt(0) below (the compiler inserts no waits for loads it did not issue).
    auto issue = [&]() {
      const int jj = max(min(jn, end - 1), 0);
      asm volatile("global_load_dword %0, %1, off" : "=v"(c_raw) : "v"(indices + jj) : "memory");
      if (vals) asm volatile("global_load_dword %0, %1, off" : "=v"(v_raw) : "v"(vals + jj) : "memory");
      else v_raw = 1.0f;
    };
    gather8_tail<false>(min(nr, 8), cs, v, sub16, X, xx, acc);
    if (nr > 8) {
      gather8_tail<true, true>(min(nr, 16) - 8, cs, v, sub16, X, xx, acc, issue);
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(c_raw), "+v"(v_raw) :: "memory");      // first use of the prefetched pair
      v = jn < end ? v_raw : 0.f;
      cs = (v == 0.f) ? 0x80000000u : (unsigned)c_raw * (unsigned)(LPR * 16);
    }
  };
  auto coop_rounds = [&](int rem) { return INTER ? min(16, (rem + G - 1) / G) : rem; };
  // index of this lane's entry in a cooperative chunk starting at `base` / in chunk q of a short row (>= end: none)
  auto coop_at = [&](int base) {
    if (!INTER) return base + 16 * g + e16;
    const int slot = lane_slot(coop_rounds(e - base));
    return slot < 0 ? e : base + G * slot + g;
  };
  auto short_at = [&](int q, int maxlen_) {
    const int slot = lane_slot(maxlen_ - 16 * q);
    return slot < 0 ? e : s + 16 * q + slot;
  };
  auto total = [&]() { return COLMASK ? accm : make_float4(acc.lo.x, acc.lo.y, acc.hi.x, acc.hi.y); };

  if (kind == 0) {
    row = __builtin_amdgcn_readfirstlane(row); s = __builtin_amdgcn_readfirstlane(s); e = __builtin_amdgcn_readfirstlane(e);
    if (ep.row_mark && ep.row_mark[row] != stamp) { leave(); return; }
    const float r = ep.row_scale ? ep.row_scale[row] : 1.0f;
    fetch(coop_at(s), e, cs, v);
    if (LATE) {
      for (int base = s; base < e; base += CH) chunk_late(coop_rounds(e - base), coop_at(base + CH), e);
    } else
    for (int base = s; base < e; base += CH) {
      const bool more = base + CH < e;
      if (prefetch && more) fetch(coop_at(base + CH), e, csn, vn);
      chunk(coop_rounds(e - base));
      if (prefetch) { cs = csn; v = vn; }
      else if (more) fetch(coop_at(base + CH), e, cs, v);
    }
    float4 a4 = total();
    if constexpr (FLOOR) {
      if (a4.x == 123456.789f) Y[0] = a4;                    // (never: keeps the sums observable)
      leave();
      return;
    }
#pragma unroll
    for (int m = LPR; m < 64; m <<= 1) a4 = f4_add(a4, f4_shfl_xor(a4, m));
    if (slot < 0) {
      row_epilogue<LPR, ADAM>(a4, row, sub, g == 0, Y, ep, r);
      leave();
      return;
    }
    if (g == 0) store_f4_sc1(partial + (size_t)slot * LPR + sub, a4);
    const int hid = __builtin_amdgcn_readfirstlane(slot_owner[slot]);
    const Heavy h = heavy[hid];
    const int hfirst = __builtin_amdgcn_readfirstlane(h.first_slot);
    const int hn = __builtin_amdgcn_readfirstlane(h.n_slots);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's write-through stores have landed
    int ticket = 0;
    if (lane == 0) ticket = __hip_atomic_fetch_add(tickets + hid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    if (ticket != hn - 1) { leave(); return; }
    if (lane == 0) __hip_atomic_store(tickets + hid, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
    float4 sum = sum_partials_agent(partial + (size_t)hfirst * LPR + sub, g, G, hn, LPR);
#pragma unroll
    for (int m = LPR; m < 64; m <<= 1) sum = f4_add(sum, f4_shfl_xor(sum, m));
    row_epilogue<LPR, ADAM>(sum, row, sub, g == 0, Y, ep, r);
    leave();
    return;
  }

  // ---- one short row per row-group ----
  const bool live = g < count && (!ep.row_mark || ep.row_mark[row] == stamp);
  // row-masked launches (last forward layer): 92 % of the nodes are dead, and a wave of four dead rows used to run on
  // through the scale load and the whole epilogue -- 3 us of a wave slot each (profiles/r02_i_*): leave at once
  if (ep.row_mark && __ballot(live) == 0ull) { leave(); return; }
  const float r = ep.row_scale ? ep.row_scale[row] : 1.0f;
  if (!live) e = s;
  int maxlen = e - s;
#pragma unroll
  for (int m = LPR; m < 64; m <<= 1) maxlen = max(maxlen, __shfl_xor(maxlen, m));
  maxlen = __builtin_amdgcn_readfirstlane(maxlen);
  fetch(short_at(0, maxlen), e, cs, v);
  if (LATE) {
    for (int q = 0; q * 16 < maxlen; ++q) chunk_late(maxlen - 16 * q, short_at(q + 1, maxlen), e);
  } else
  for (int q = 0; q * 16 < maxlen; ++q) {
    const bool more = (q + 1) * 16 < maxlen;
    if (prefetch && more) fetch(short_at(q + 1, maxlen), e, csn, vn);
    chunk(maxlen - 16 * q);
    if (prefetch) { cs = csn; v = vn; }
    else if (more) fetch(short_at(q + 1, maxlen), e, cs, v);
  }
  if constexpr (FLOOR) {
    const float4 a4 = total();
    if (a4.x == 123456.789f) Y[0] = a4;
  } else {
    row_epilogue<LPR, ADAM>(total(), row, sub, live, Y, ep, r);
  }
  leave();
}

// ---------------------------------------------------------------------------------------------
// Three value arrays over one structure, one input: y_v = A_v x for v = 0..2 in ONE traversal.  SGL's
// first layer multiplies the full adjacency and its two edge-dropped views (same indptr / indices,
// dropped entries are zeros: data/augmentor.py:29-40) by the same ego table, so the x rows -- the
// traffic that bounds this kernel -- are gathered once for all three.  Same schedule, DPP broadcasts
// and in-kernel finish of split rows as spmm_rows_kernel; no epilogue (the layer has none).
// ---------------------------------------------------------------------------------------------
template <int LPR, int T0>
__device__ __forceinline__ void gather8x3(int c, float v0, float v1, float v2, const float4* __restrict__ X, int sub,
                                          float4 (&acc)[3]) {
  int cc[8];
  float a0[8], a1[8], a2[8];
  float4 xx[8];
#define SRH_BC(T) cc[T] = row_bcast_i<T0 + T>(c); a0[T] = row_bcast_f<T0 + T>(v0); a1[T] = row_bcast_f<T0 + T>(v1); \
  a2[T] = row_bcast_f<T0 + T>(v2);
  SRH_BC(0) SRH_BC(1) SRH_BC(2) SRH_BC(3) SRH_BC(4) SRH_BC(5) SRH_BC(6) SRH_BC(7)
#undef SRH_BC
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    xx[t] = f4_zero();
    if (a0[t] != 0.f) xx[t] = ld_x<LPR>(X, cc[t], sub);     // (views are sub-graphs: a0 == 0 only for padding)
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    acc[0] = f4_fma(a0[t], xx[t], acc[0]);
    acc[1] = f4_fma(a1[t], xx[t], acc[1]);
    acc[2] = f4_fma(a2[t], xx[t], acc[2]);
  }
}

template <int LPR>
__global__ __launch_bounds__(256) void spmm_rows3_kernel(const Task* __restrict__ tasks, int n_tasks,
                                                         const Seg* __restrict__ segs,
                                                         const int32_t* __restrict__ indices,
                                                         const float* __restrict__ vals0, const float* __restrict__ vals1,
                                                         const float* __restrict__ vals2, const float4* __restrict__ X,
                                                         float4* __restrict__ Y0, float4* __restrict__ Y1,
                                                         float4* __restrict__ Y2, float4* __restrict__ partial,
                                                         const Heavy* __restrict__ heavy,
                                                         const int32_t* __restrict__ slot_owner,
                                                         int32_t* __restrict__ tickets) {
  static_assert(3 * LPR <= 64, "three partial rows must fit one 256-float partial slot");
  constexpr int G = 64 / LPR;
  constexpr int CH = 16 * G;
  // (wave-uniform by construction: readfirstlane lets the compiler fetch the task record with a scalar load)
  const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256u + threadIdx.x) >> 6));
  if (wave >= n_tasks) return;
  const int lane = threadIdx.x & 63;
  const int g = lane / LPR, sub = lane % LPR, e16 = lane & 15;
  const Task tk = tasks[wave];
  const int kind = __builtin_amdgcn_readfirstlane(tk.kind);
  const int first = __builtin_amdgcn_readfirstlane(tk.first);
  const int count = __builtin_amdgcn_readfirstlane(tk.count);
  float4 acc[3] = {f4_zero(), f4_zero(), f4_zero()};
  float4* const Y[3] = {Y0, Y1, Y2};

  if (kind == 0) {
    const Seg sg = segs[first];
    const int row = __builtin_amdgcn_readfirstlane(sg.row), s = __builtin_amdgcn_readfirstlane(sg.start);
    const int e = __builtin_amdgcn_readfirstlane(sg.end), slot = __builtin_amdgcn_readfirstlane(sg.slot);
    for (int base = s; base < e; base += CH) {
      const int j = base + G * e16 + g;                  // (dealt round-robin to the groups, as spmm_rows_kernel does)
      int c = 0;
      float v0 = 0.f, v1 = 0.f, v2 = 0.f;
      if (j < e) { c = indices[j]; v0 = vals0[j]; v1 = vals1[j]; v2 = vals2[j]; }
      gather8x3<LPR, 0>(c, v0, v1, v2, X, sub, acc);
      if (e - base > 8) gather8x3<LPR, 8>(c, v0, v1, v2, X, sub, acc);
    }
#pragma unroll
    for (int v = 0; v < 3; ++v)
#pragma unroll
      for (int m = LPR; m < 64; m <<= 1) acc[v] = f4_add(acc[v], f4_shfl_xor(acc[v], m));
    if (slot < 0) {
      if (g == 0) {
#pragma unroll
        for (int v = 0; v < 3; ++v) Y[v][(size_t)row * LPR + sub] = acc[v];
      }
      return;
    }
    if (g == 0) {
#pragma unroll
      for (int v = 0; v < 3; ++v) store_f4_sc1(partial + (size_t)slot * 64 + v * LPR + sub, acc[v]);
    }
    const int hid = __builtin_amdgcn_readfirstlane(slot_owner[slot]);
    const Heavy h = heavy[hid];
    const int hfirst = __builtin_amdgcn_readfirstlane(h.first_slot);
    const int hn = __builtin_amdgcn_readfirstlane(h.n_slots);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int ticket = 0;
    if (lane == 0) ticket = __hip_atomic_fetch_add(tickets + hid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    if (ticket != hn - 1) return;
    if (lane == 0) __hip_atomic_store(tickets + hid, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      float4 sum = sum_partials_agent(partial + (size_t)hfirst * 64 + v * LPR + sub, g, G, hn, 64);
#pragma unroll
      for (int m = LPR; m < 64; m <<= 1) sum = f4_add(sum, f4_shfl_xor(sum, m));
      if (g == 0) Y[v][(size_t)row * LPR + sub] = sum;
    }
    return;
  }

  // ---- one short row per row-group ----
  const bool have = g < count;
  const Seg sg = segs[first + (have ? g : 0)];
  const int row = sg.row, s = sg.start;
  const int e = have ? sg.end : s;
  int maxlen = e - s;
#pragma unroll
  for (int m = LPR; m < 64; m <<= 1) maxlen = max(maxlen, __shfl_xor(maxlen, m));
  maxlen = __builtin_amdgcn_readfirstlane(maxlen);
  for (int q = 0; q * 16 < maxlen; ++q) {
    const int j = s + 16 * q + e16;
    int c = 0;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
    if (j < e) { c = indices[j]; v0 = vals0[j]; v1 = vals1[j]; v2 = vals2[j]; }
    gather8x3<LPR, 0>(c, v0, v1, v2, X, sub, acc);
    if (maxlen - 16 * q > 8) gather8x3<LPR, 8>(c, v0, v1, v2, X, sub, acc);
  }
  if (have) {
#pragma unroll
    for (int v = 0; v < 3; ++v) Y[v][(size_t)row * LPR + sub] = acc[v];
  }
}

// ---------------------------------------------------------------------------------------------
// Column slices: the column-sharded multi-GPU layout keeps DL = d / G columns of every
// (N, d) table on each rank, so a propagation layer needs no exchange at all -- the product is independent per
// column -- and a gathered x row is only DL * 4 = 32 .. 128 bytes.  The helpers below serve the 8-column kernel
// (spmm_pair_kernel): vector loads / stores of EPL = DL / 8 floats, the perturbation and the epilogue on a lane
// that ends up holding EPL columns of a finished row.
// The PERTURB unit vector is normalised over the whole d-wide row (XSimGCL.py:90): with the counter RNG every
// rank regenerates the row's other columns (hash only, no memory), with injected noise it reads the full noise
// row; its own columns use exactly the counters of the one-GPU kernel, so a sharded run sees the same
// perturbation as an unsharded one.
// ---------------------------------------------------------------------------------------------
template <int EPL> struct ThinVec;
template <> struct ThinVec<1> { using type = float; };
template <> struct ThinVec<2> { using type = float2; };
template <> struct ThinVec<4> { using type = float4; };

template <int EPL>
__device__ __forceinline__ void ld_epl(const float* p, float (&v)[EPL]) {
  using V = typename ThinVec<EPL>::type;
  union { V vec; float f[EPL]; } u;
  u.vec = *reinterpret_cast<const V*>(p);
#pragma unroll
  for (int i = 0; i < EPL; ++i) v[i] = u.f[i];
}
template <int EPL>
__device__ __forceinline__ void st_epl(float* p, const float (&v)[EPL]) {
  using V = typename ThinVec<EPL>::type;
  union { V vec; float f[EPL]; } u;
#pragma unroll
  for (int i = 0; i < EPL; ++i) u.f[i] = v[i];
  *reinterpret_cast<V*>(p) = u.vec;
}
// write-through (sc1) store of a partial: not left dirty in this XCD's L2 (see store_f4_sc1)
template <int EPL>
__device__ __forceinline__ void st_epl_sc1(float* p, const float (&v)[EPL]) {
  if constexpr (EPL == 1) {
    asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v[0]) : "memory");
  } else if constexpr (EPL == 2) {
    typedef float floatx2_t __attribute__((ext_vector_type(2)));
    floatx2_t x = {v[0], v[1]};
    asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(x) : "memory");
  } else {
    floatx4_t x = {v[0], v[1], v[2], v[3]};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
  }
}
template <int EPL>
__device__ __forceinline__ void ld_epl_agent(const float* p, float (&v)[EPL]) {
#pragma unroll
  for (int i = 0; i < EPL; ++i) v[i] = __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ float pick4(uint4 r, int comp) {
  const uint32_t w = (comp == 0) ? r.x : (comp == 1) ? r.y : (comp == 2) ? r.z : r.w;
  return u01(w);
}

template <int DL>
__device__ __forceinline__ void thin_perturb(float (&y)[DL / 8], const float (&raw)[DL / 8], int row, int el,
                                             const float* noise, uint32_t off_lo, uint32_t off_hi,
                                             const DevEpilogue& ep) {
  constexpr int EPL = DL / 8;
  const int dfull = ep.noise_d_full;
  const int e0 = ep.noise_col0 + el * EPL;          // first of this lane's columns in the whole row
  float nu[EPL];
  float ss = 0.f;
  if (noise) {
    const float* nr = noise + (size_t)row * dfull;
    const int per = dfull / 8;                      // this lane's share of the row for the norm
    for (int t = 0; t < per; t += 4) {
      const float4 z = *reinterpret_cast<const float4*>(nr + el * per + t);
      ss += f4_dot(z, z);
    }
#pragma unroll
    for (int i = 0; i < EPL; ++i) nu[i] = nr[e0 + i];
  } else {
    uint64_t ctr = (((uint64_t)off_hi << 32) | off_lo) + (uint64_t)row;
    if (ep.rng_step) ctr += (uint64_t)(*ep.rng_step) * ep.rng_stride;
    for (int sub = el; sub < dfull / 4; sub += 8) {
      const uint4 r = counter_rng4(ctr, (uint32_t)sub, ep.seed_lo, ep.seed_hi);
      const float4 z = make_float4(u01(r.x), u01(r.y), u01(r.z), u01(r.w));
      ss += f4_dot(z, z);
    }
    const uint4 r = counter_rng4(ctr, (uint32_t)(e0 >> 2), ep.seed_lo, ep.seed_hi);
#pragma unroll
    for (int i = 0; i < EPL; ++i) nu[i] = pick4(r, (e0 & 3) + i);
  }
  ss = group_sum<8>(ss);
  const float scale = ep.eps / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
  for (int i = 0; i < EPL; ++i) y[i] = raw[i] + sgnf(raw[i]) * (nu[i] * scale);
}

template <int DL>
__device__ __forceinline__ void thin_epilogue(float (&y)[DL / 8], int row, int el, bool store, float* __restrict__ Y,
                                              const DevEpilogue& ep) {
  // el: which EPL columns of the slice this lane holds -- any bijection of 0..7 over the 8 lanes of a group
  constexpr int EPL = DL / 8;
  const size_t at = (size_t)row * DL + el * EPL;
  if (ep.flags & SRH_EPI_AXPY) {
#pragma unroll
    for (int i = 0; i < EPL; ++i) y[i] *= ep.alpha;
    const bool marked = !ep.add_mark || ep.add_mark[row] == (int)(*ep.mark_stamp);
    for (int t = 0; t < ep.n_add; ++t) {
      if (((ep.add_sparse >> t) & 1) && !marked) continue;
      float a[EPL];
      ld_epl<EPL>(ep.add[t] + at, a);
#pragma unroll
      for (int i = 0; i < EPL; ++i) y[i] = fmaf(ep.add_scale[t], a[i], y[i]);
    }
  }
  if (ep.flags & SRH_EPI_PERTURB) {
    float raw[EPL];
#pragma unroll
    for (int i = 0; i < EPL; ++i) raw[i] = y[i];
    if (!ep.main_clean) thin_perturb<DL>(y, raw, row, el, ep.noise, ep.off_lo, ep.off_hi, ep);
    for (int k = 0; k < ep.n_extra; ++k) {
      float yk[EPL];
      thin_perturb<DL>(yk, raw, row, el, ep.extra_noise[k], ep.extra_off_lo[k], ep.extra_off_hi[k], ep);
      if (store) st_epl<EPL>(ep.extra_out[k] + at, yk);
    }
  }
  if (store) st_epl<EPL>(Y + at, y);
  if (ep.flags & SRH_EPI_MEAN) {
    float m[EPL];
#pragma unroll
    for (int i = 0; i < EPL; ++i) m[i] = 0.f;
    for (int t = 0; t < ep.n_prev; ++t) {
      float a[EPL];
      ld_epl<EPL>(ep.prev[t] + at, a);
#pragma unroll
      for (int i = 0; i < EPL; ++i) m[i] = (t == 0) ? a[i] : m[i] + a[i];
    }
#pragma unroll
    for (int i = 0; i < EPL; ++i) m[i] = (ep.n_prev > 0 ? m[i] + y[i] : y[i]) * ep.mean_rcp;
    if (store) st_epl<EPL>(ep.mean_out + at, m);
  }
}

// ---------------------------------------------------------------------------------------------
// 8-column slices, two lanes per x row.  Where the time of a lane-per-row kernel went was measured by
// knocking parts out (Yelp2018 shape, 23.5 us): without the x gathers 11.0, without the (col, val) loads
// 21.1, without both 10.2 -- the gathers cost 12.5 us, and they cost it per L1 LOOK-UP (a lane that owns
// a 32-byte row issues two 16-byte loads = two look-ups of the same line), not per byte.  Here the two
// lanes of a pair fetch the two halves of a row with ONE instruction -- one look-up per entry -- and
// 4 pairs share a short row (entries p, p + 4, ...), 32 pairs a long one; a lane keeps 4 partial sums,
// and a 2-step butterfly over the 4 pairs (xor 4, xor 2) leaves each lane one column of the row.
// 60 VGPRs instead of 90: 8 waves per SIMD.
// ---------------------------------------------------------------------------------------------
// one task of spmm_pair_kernel: tk / sg are this lane's copies of the task record and of its segment
// (kind 0: the wave's one segment; kind 1: the short row of the lane's 8-lane group)
__device__ __forceinline__ void pair_task(const Task& tk, const Seg& sgl, const int32_t* __restrict__ indices,
                                          const float* __restrict__ vals, const float* __restrict__ X,
                                          float* __restrict__ Y, float* __restrict__ partial,
                                          const Heavy* __restrict__ heavy, const int32_t* __restrict__ slot_owner,
                                          int32_t* __restrict__ tickets, const DevEpilogue& ep, int stamp, int lane) {
  constexpr int DL = 8;
  const int g = lane >> 3, e8 = lane & 7, h = lane & 1, pg = e8 >> 1;     // group, lane in group, half, pair in group
  const bool b2 = (e8 & 4) != 0, b1 = (e8 & 2) != 0;
  const int el = h * 4 + (b2 ? 2 : 0) + (b1 ? 1 : 0);                      // the column this lane ends up with
  const int kind = __builtin_amdgcn_readfirstlane(tk.kind);
  const int count = __builtin_amdgcn_readfirstlane(tk.count);
  const float4* Xh = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(X) + h * 16);
  float4 acc = f4_zero();

  // 8 entries of this pair: j, j + stride, ...
  auto accumulate = [&](int j, int stride, int e) {
    int c[8];
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int jj = j + k * stride;
      c[k] = 0;
      v[k] = 0.f;
      if (jj < e) { c[k] = indices[jj]; v[k] = vals[jj]; }
    }
    if (ep.col_mark) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (v[k] != 0.f && ep.col_mark[c[k]] != stamp) v[k] = 0.f;
    }
    float4 x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      x[k] = f4_zero();
      if (v[k] != 0.f)
        x[k] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(Xh) + (unsigned)c[k] * 32u);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) acc = f4_fma(v[k], x[k], acc);
  };
  // sum over the 4 pairs of a group; afterwards lane (h, b2, b1) holds column h*4 + 2*b2 + b1
  auto reduce4 = [&]() {
    const float k0 = b2 ? acc.z : acc.x, k1 = b2 ? acc.w : acc.y;
    const float s0 = b2 ? acc.x : acc.z, s1 = b2 ? acc.y : acc.w;
    const float a0 = k0 + __shfl_xor(s0, 4), a1 = k1 + __shfl_xor(s1, 4);
    const float keep = b1 ? a1 : a0, send = b1 ? a0 : a1;
    return keep + __shfl_xor(send, 2);
  };
  float out[1];

  if (kind == 0) {
    const int row = __builtin_amdgcn_readfirstlane(sgl.row), s = __builtin_amdgcn_readfirstlane(sgl.start);
    const int e = __builtin_amdgcn_readfirstlane(sgl.end), slot = __builtin_amdgcn_readfirstlane(sgl.slot);
    if (ep.row_mark && ep.row_mark[row] != stamp) return;
    for (int base = s; base < e; base += 256) accumulate(base + (lane >> 1), 32, e);
    out[0] = reduce4();
    out[0] += __shfl_xor(out[0], 8);
    out[0] += __shfl_xor(out[0], 16);
    out[0] += __shfl_xor(out[0], 32);
    if (slot < 0) {
      thin_epilogue<DL>(out, row, el, g == 0, Y, ep);
      return;
    }
    if (g == 0) st_epl_sc1<1>(partial + (size_t)slot * DL + el, out);
    const int hid = __builtin_amdgcn_readfirstlane(slot_owner[slot]);
    const Heavy hv = heavy[hid];
    const int hfirst = __builtin_amdgcn_readfirstlane(hv.first_slot);
    const int hn = __builtin_amdgcn_readfirstlane(hv.n_slots);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's write-through stores have landed
    int ticket = 0;
    if (lane == 0) ticket = __hip_atomic_fetch_add(tickets + hid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    if (ticket != hn - 1) return;
    if (lane == 0) __hip_atomic_store(tickets + hid, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
    float sum[1] = {0.f};
    for (int t = g; t < hn; t += 8) {
      float pz[1];
      ld_epl_agent<1>(partial + (size_t)(hfirst + t) * DL + el, pz);
      sum[0] += pz[0];
    }
    sum[0] += __shfl_xor(sum[0], 8);
    sum[0] += __shfl_xor(sum[0], 16);
    sum[0] += __shfl_xor(sum[0], 32);
    thin_epilogue<DL>(sum, row, el, g == 0, Y, ep);
    return;
  }

  // ---- one short row per 8-lane group, 4 pairs striding over its entries ----
  const bool have = g < count;
  const int row = sgl.row, s = sgl.start;
  const bool live = have && (!ep.row_mark || ep.row_mark[row] == stamp);
  const int e = live ? sgl.end : s;
  for (int base = s; __any(base < e); base += 32) accumulate(base + pg, 4, e);
  out[0] = reduce4();
  thin_epilogue<DL>(out, row, el, live, Y, ep);
}

__global__ __launch_bounds__(256) void spmm_pair_kernel(const Task* __restrict__ tasks, int n_tasks,
                                                        const Seg* __restrict__ segs,
                                                        const int32_t* __restrict__ indices,
                                                        const float* __restrict__ vals, const float* __restrict__ X,
                                                        float* __restrict__ Y, float* __restrict__ partial,
                                                        const Heavy* __restrict__ heavy,
                                                        const int32_t* __restrict__ slot_owner,
                                                        int32_t* __restrict__ tickets, DevEpilogue ep) {
  // (one task per wave: giving a wave 2 / 4 tasks with their records fetched up front -- half / a quarter of the
  // waves, two round trips saved per extra task -- measured 22.9 / 22.8 us against 19.1: this launch wants MORE
  // waves in flight, not fewer; workgroups of 64 .. 1024 threads: 19.1 / 19.2 / 19.1 / 20.5 / 20.8 us)
  // (wave-uniform by construction: readfirstlane lets the compiler fetch the task record with a scalar load)
  const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256u + threadIdx.x) >> 6));
  if (wave >= n_tasks) return;
  const int lane = threadIdx.x & 63;
  const int stamp = ep.mark_stamp ? (int)(*ep.mark_stamp) : 0;
  const Task tk = tasks[wave];
  const Seg sg = segs[tk.first + ((tk.kind == 1 && (lane >> 3) < tk.count) ? (lane >> 3) : 0)];
  pair_task(tk, sg, indices, vals, X, Y, partial, heavy, slot_owner, tickets, ep, stamp, lane);
}

// ---------------------------------------------------------------------------------------------
// 16- and 32-column slices (column-sharded layout at G = 4 / 2 for d = 64): spmm_rows_kernel's
// schedule with LPR = 4 / 8 lanes per row, so that ONE gather instruction fetches a whole 64 / 128-byte
// x row per row-group -- one L1 lookup per entry (a lane-per-entry mapping pays
// one per 16 bytes: measured 39 / 67 us per Yelp-shape launch at 16 / 32 columns against 26 us at 8).
//   * a row-group consumes 8 entries per iteration, held by its own lanes as 8 / LPR sub-blocks of LPR
//     consecutive entries; entry t of a sub-block is broadcast inside the group with DPP -- quad_perm
//     for LPR = 4 (a group is a quad), two row_newbcast + a select for LPR = 8 (two groups per DPP row);
//   * 8 gathers are in flight per row-group before the first FMA; G = 64 / LPR = 16 / 8 short rows per
//     wave (task lists of the plan), or all groups on one long row / split segment (8 G entries per
//     iteration) summed with xor-shuffles; split rows finish in-kernel (write-through partials + ticket).
// ---------------------------------------------------------------------------------------------
template <int LPR, int T>
__device__ __forceinline__ int grp_bcast_i(int x, int lane) {
  if constexpr (LPR == 4) {
    return __builtin_amdgcn_update_dpp(0, x, T * 0x55, 0xf, 0xf, false);           // quad_perm [T,T,T,T]
  } else {
    const int lo = __builtin_amdgcn_update_dpp(0, x, 0x150 + T, 0xf, 0xf, false);      // row_newbcast:T
    const int hi = __builtin_amdgcn_update_dpp(0, x, 0x150 + T + 8, 0xf, 0xf, false);  // row_newbcast:T+8
    return (lane & 8) ? hi : lo;
  }
}
template <int LPR, int T>
__device__ __forceinline__ float grp_bcast_f(float x, int lane) {
  return __int_as_float(grp_bcast_i<LPR, T>(__float_as_int(x), lane));
}

// 8 entries of this row-group: sub-block b holds entries b*LPR .. b*LPR+LPR-1 in (c[b], v[b]) of its lanes
template <int LPR>
__device__ __forceinline__ void slice_gather8(const int (&c)[8 / LPR], const float (&v)[8 / LPR],
                                              const float4* __restrict__ X, int sub, int lane, float4& acc) {
  int cc[8];
  float vv[8];
  float4 xx[8];
  if constexpr (LPR == 8) {
#define SRH_BC(T) cc[T] = grp_bcast_i<8, T>(c[0], lane); vv[T] = grp_bcast_f<8, T>(v[0], lane);
    SRH_BC(0) SRH_BC(1) SRH_BC(2) SRH_BC(3) SRH_BC(4) SRH_BC(5) SRH_BC(6) SRH_BC(7)
#undef SRH_BC
  } else {
#define SRH_BC(T) cc[T] = grp_bcast_i<4, T>(c[0], lane); vv[T] = grp_bcast_f<4, T>(v[0], lane); \
                  cc[4 + T] = grp_bcast_i<4, T>(c[1], lane); vv[4 + T] = grp_bcast_f<4, T>(v[1], lane);
    SRH_BC(0) SRH_BC(1) SRH_BC(2) SRH_BC(3)
#undef SRH_BC
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    xx[t] = f4_zero();
    if (vv[t] != 0.f) xx[t] = ld_x<LPR>(X, cc[t], sub);     // padding / dropped / dead columns: no gather
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) acc = f4_fma(vv[t], xx[t], acc);
}

template <int LPR>
__global__ __launch_bounds__(256) void spmm_slice_kernel(const Task* __restrict__ tasks, int n_tasks,
                                                         const Seg* __restrict__ segs,
                                                         const int32_t* __restrict__ indices,
                                                         const float* __restrict__ vals,
                                                         const float4* __restrict__ X, float4* __restrict__ Y,
                                                         float4* __restrict__ partial,
                                                         const Heavy* __restrict__ heavy,
                                                         const int32_t* __restrict__ slot_owner,
                                                         int32_t* __restrict__ tickets, DevEpilogue ep) {
  static_assert(LPR == 4 || LPR == 8, "16- and 32-column slices");
  constexpr int G = 64 / LPR;          // row-groups per wave
  constexpr int NB = 8 / LPR;          // sub-blocks of LPR entries a group holds per iteration
  // (wave-uniform by construction: readfirstlane lets the compiler fetch the task record with a scalar load)
  const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256u + threadIdx.x) >> 6));
  if (wave >= n_tasks) return;
  const int lane = threadIdx.x & 63;
  const int g = lane / LPR, sub = lane % LPR;
  const Task tk = tasks[wave];
  const int kind = __builtin_amdgcn_readfirstlane(tk.kind);
  const int first = __builtin_amdgcn_readfirstlane(tk.first);
  const int count = __builtin_amdgcn_readfirstlane(tk.count);
  const int stamp = ep.mark_stamp ? (int)(*ep.mark_stamp) : 0;
  float4 acc = f4_zero();

  if (kind == 0) {
    const Seg sg = segs[first];
    const int row = __builtin_amdgcn_readfirstlane(sg.row), s = __builtin_amdgcn_readfirstlane(sg.start);
    const int e = __builtin_amdgcn_readfirstlane(sg.end), slot = __builtin_amdgcn_readfirstlane(sg.slot);
    if (ep.row_mark && ep.row_mark[row] != stamp) return;
    // the (col, val) pairs of 512 entries are loaded up front -- one index latency per segment instead of one
    // per 8 G entries: with a few thousand waves per launch this kernel runs on its latency chain, not on bytes
    constexpr int KCH = 512 / (8 * G);       // iterations per 512 entries: 8 (LPR = 8), 4 (LPR = 4)
    for (int base = s; base < e; base += 512) {
      int cq[KCH][NB];
      float vq[KCH][NB];
#pragma unroll
      for (int k = 0; k < KCH; ++k) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const int j = base + 8 * G * k + 8 * g + b * LPR + sub;
          cq[k][b] = 0;
          vq[k][b] = 0.f;
          if (j < e) { cq[k][b] = indices[j]; vq[k][b] = vals[j]; }
        }
      }
      if (ep.col_mark) {
#pragma unroll
        for (int k = 0; k < KCH; ++k)
#pragma unroll
          for (int b = 0; b < NB; ++b)
            if (vq[k][b] != 0.f && ep.col_mark[cq[k][b]] != stamp) vq[k][b] = 0.f;
      }
#pragma unroll
      for (int k = 0; k < KCH; ++k)
        if (base + 8 * G * k < e) slice_gather8<LPR>(cq[k], vq[k], X, sub, lane, acc);
    }
#pragma unroll
    for (int m = LPR; m < 64; m <<= 1) acc = f4_add(acc, f4_shfl_xor(acc, m));
    if (slot < 0) {
      row_epilogue<LPR>(acc, row, sub, g == 0, Y, ep);
      return;
    }
    if (g == 0) store_f4_sc1(partial + (size_t)slot * LPR + sub, acc);
    const int hid = __builtin_amdgcn_readfirstlane(slot_owner[slot]);
    const Heavy h = heavy[hid];
    const int hfirst = __builtin_amdgcn_readfirstlane(h.first_slot);
    const int hn = __builtin_amdgcn_readfirstlane(h.n_slots);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's write-through stores have landed
    int ticket = 0;
    if (lane == 0) ticket = __hip_atomic_fetch_add(tickets + hid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    if (ticket != hn - 1) return;
    if (lane == 0) __hip_atomic_store(tickets + hid, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
    float4 sum = sum_partials_agent(partial + (size_t)hfirst * LPR + sub, g, G, hn, LPR);
#pragma unroll
    for (int m = LPR; m < 64; m <<= 1) sum = f4_add(sum, f4_shfl_xor(sum, m));
    row_epilogue<LPR>(sum, row, sub, g == 0, Y, ep);
    return;
  }

  // ---- one short row per row-group ----
  const bool have = g < count;
  const Seg sg = segs[first + (have ? g : 0)];
  const int row = sg.row, s = sg.start;
  const bool live = have && (!ep.row_mark || ep.row_mark[row] == stamp);
  const int e = live ? sg.end : s;
  int maxlen = e - s;
#pragma unroll
  for (int m = LPR; m < 64; m <<= 1) maxlen = max(maxlen, __shfl_xor(maxlen, m));
  maxlen = __builtin_amdgcn_readfirstlane(maxlen);
  for (int q0 = 0; q0 * 8 < maxlen; q0 += 8) {        // (one pass: short rows have at most 64 entries)
    int cq[8][NB];
    float vq[8][NB];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int j = s + 8 * (q0 + k) + b * LPR + sub;
        cq[k][b] = 0;
        vq[k][b] = 0.f;
        if (j < e) { cq[k][b] = indices[j]; vq[k][b] = vals[j]; }
      }
    }
    if (ep.col_mark) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int b = 0; b < NB; ++b)
          if (vq[k][b] != 0.f && ep.col_mark[cq[k][b]] != stamp) vq[k][b] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if ((q0 + k) * 8 < maxlen) slice_gather8<LPR>(cq[k], vq[k], X, sub, lane, acc);
  }
  row_epilogue<LPR>(acc, row, sub, live, Y, ep);
}

// ---------------------------------------------------------------------------------------------
// srh_gather_floor_probe: the bare gather stream of a product -- nothing else.  Every wave walks a strided share of an index
// array 8 * G entries at a time (the next iteration's indices in flight under the current gathers), fetches the 4 * LPR-float
// x row of each entry with 8 loads in flight per row-group and adds it into a register; no values, no row structure, no
// epilogue, no y.  Run over the LIVE graph's CSR column array it times what the vector-memory path of this chip needs for the
// row fetches one propagation launch makes: the floor bench.py prints as roofline.gather_floor_us next to the launch's own
// time (the figure used to live in tools/microbench/gather_zipf.hip on a synthetic Zipf stream).
// ---------------------------------------------------------------------------------------------
template <int LPR>
__global__ __launch_bounds__(256) void gather_floor_kernel(const float4* __restrict__ X, const int32_t* __restrict__ idx,
                                                           long n_idx, float4* __restrict__ sink) {
  constexpr int G = 64 / LPR;
  const int lane = threadIdx.x & 63, g = lane / LPR, sub = lane % LPR;
  const long wave = (blockIdx.x * 256L + threadIdx.x) >> 6, n_waves = gridDim.x * 4L;
  float4 acc = f4_zero();
  long base = wave * 8 * G;
  int nxt[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) nxt[t] = (base + 8 * G <= n_idx) ? idx[base + t * G + g] : 0;
  for (; base + 8 * G <= n_idx; base += n_waves * 8 * G) {
    int cur[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) cur[t] = nxt[t];
    const long nb = base + n_waves * 8 * G;
    if (nb + 8 * G <= n_idx) {
#pragma unroll
      for (int t = 0; t < 8; ++t) nxt[t] = idx[nb + t * G + g];
    }
    float4 x[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = ld_x<LPR>(X, cur[t], sub);
#pragma unroll
    for (int t = 0; t < 8; ++t) acc = f4_add(acc, x[t]);
  }
  if (acc.x == 123.456f) sink[0] = acc;       // (never true on finite tables: keeps the loads alive)
}

}  // namespace

struct srh_spmm_plan {
  int64_t n_rows = 0, n_cols = 0, nnz = 0;
  int32_t n_heavy = 0, n_slots = 0, split_len = 0;
  // one task per wave; one task list per row-group count G = 64 / LPR: index 0..3 = 8, 4, 2, 1 rows per wave
  // (LPR = 8, 16, 32, 64), index 4 = 16 rows per wave (LPR = 4: 16-column slices)
  int32_t n_tasks[5] = {0, 0, 0, 0, 0};
  Task* d_tasks[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  Seg* d_tsegs = nullptr;          // segments in task order (+ 16 padding records: a short-row task may read past its count)
  Task64* d_tasks64[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // spmm_rows_kernel's records (index 1..3)
  // srh_spmm_plan_set_xcd_shares: the list a launch runs may deal the XCDs unequal numbers of blocks (empty records
  // pad the shorter queues); the canonical list stays on the host
  std::vector<Task64> h_tasks64[5];
  int32_t n_run[5] = {0, 0, 0, 0, 0};       // records in d_tasks64[gi] (a multiple of 32 once shares are set)
  size_t cap64[5] = {0, 0, 0, 0, 0};        // records d_tasks64[gi] has room for
  Heavy* d_heavy = nullptr;
  int32_t* d_slot_owner = nullptr;
  int32_t* d_tickets = nullptr;    // one arrival counter per split row, self re-arming
  float* d_partial = nullptr;      // n_slots * 256 floats (enough for d <= 256)
};

// host view of the epilogue (include/selfrec_hip.h) -> the kernels' argument block, with every check the ABI promises
static srh_status_t translate_epilogue(const srh_spmm_epilogue_t* epi, int32_t d, const float* d_x, const float* d_y,
                                       DevEpilogue& ep) {
  if (epi) {
    SRH_REQUIRE((epi->flags & ~(SRH_EPI_PERTURB | SRH_EPI_MEAN | SRH_EPI_AXPY | SRH_EPI_ADAM)) == 0,
                "spmm_f32: unknown epilogue flag");
    ep.flags = epi->flags;
    ep.eps = epi->eps;
    ep.noise = epi->d_noise;
    ep.seed_lo = (uint32_t)epi->rng_seed; ep.seed_hi = (uint32_t)(epi->rng_seed >> 32);
    ep.off_lo = (uint32_t)epi->rng_offset; ep.off_hi = (uint32_t)(epi->rng_offset >> 32);
    ep.rng_step = epi->d_rng_step;
    ep.rng_stride = epi->rng_stride;
    ep.row_mark = epi->d_row_mark;
    ep.col_mark = epi->d_col_mark;
    ep.mark_stamp = epi->d_mark_stamp;
    ep.add_mark = epi->d_add_mark;
    ep.add_sparse = epi->add_sparse_mask;
    if (epi->d_row_scale || epi->scale_flags || epi->prev_unscale_mask || epi->add_rowscale_mask) {
      SRH_REQUIRE(epi->d_row_scale && d >= 64, "spmm_f32: row scaling needs d_row_scale and d >= 64");
      SRH_REQUIRE((epi->scale_flags & ~(SRH_SCALE_IN | SRH_SCALE_OUT)) == 0, "spmm_f32: unknown scale flag");
      ep.row_scale = epi->d_row_scale;
      ep.scale_flags = epi->scale_flags;
      ep.prev_unscale = epi->prev_unscale_mask;
      ep.add_rowscale = epi->add_rowscale_mask;
    }
    if (epi->noise_d_full) {       // y is a column slice of noise_d_full-wide rows (column-sharded tables)
      SRH_REQUIRE(epi->noise_d_full % 32 == 0 && epi->noise_col0 >= 0 && epi->noise_col0 % d == 0 &&
                      epi->noise_col0 + d <= epi->noise_d_full,
                  "spmm_f32: column slice [%d, %d) of %d-wide rows is not supported (aligned slices of rows a multiple of 32 wide)",
                  epi->noise_col0, epi->noise_col0 + d, epi->noise_d_full);
      ep.noise_d_full = epi->noise_d_full;
      ep.noise_col0 = epi->noise_col0;
    }
    if (epi->n_extra || epi->main_clean) {
      SRH_REQUIRE((epi->flags & SRH_EPI_PERTURB) && !(epi->flags & SRH_EPI_MEAN) && epi->n_extra >= 0 &&
                      epi->n_extra <= SRH_MAX_EXTRA,
                  "spmm_f32: extra perturbed outputs need PERTURB without MEAN and at most %d of them", SRH_MAX_EXTRA);
      ep.n_extra = epi->n_extra;
      ep.main_clean = epi->main_clean != 0;
      for (int k = 0; k < epi->n_extra; ++k) {
        SRH_REQUIRE(epi->d_extra_out[k] && epi->d_extra_out[k] != d_y && epi->d_extra_out[k] != d_x,
                    "spmm_f32: bad extra output %d", k);
        ep.extra_out[k] = epi->d_extra_out[k];
        ep.extra_noise[k] = epi->d_extra_noise[k];
        ep.extra_off_lo[k] = (uint32_t)epi->extra_rng_offset[k];
        ep.extra_off_hi[k] = (uint32_t)(epi->extra_rng_offset[k] >> 32);
      }
    }
    SRH_REQUIRE(!(ep.row_mark || ep.col_mark || ep.add_mark) || ep.mark_stamp, "spmm_f32: activity marks need d_mark_stamp");
    if (epi->flags & SRH_EPI_MEAN) {
      SRH_REQUIRE(epi->n_prev >= 0 && epi->n_prev <= SRH_MAX_PREV && epi->d_mean_out && epi->mean_div != 0.f,
                  "spmm_f32: bad MEAN epilogue");
      ep.n_prev = epi->n_prev;
      for (int t = 0; t < epi->n_prev; ++t) {
        SRH_REQUIRE(epi->d_prev[t], "spmm_f32: null prev[%d]", t);
        ep.prev[t] = epi->d_prev[t];
      }
      ep.mean_rcp = 1.0f / epi->mean_div;
      ep.mean_out = epi->d_mean_out;
    }
    if (epi->flags & SRH_EPI_AXPY) {
      SRH_REQUIRE(epi->n_add >= 0 && epi->n_add <= SRH_MAX_ADD, "spmm_f32: bad AXPY epilogue");
      ep.n_add = epi->n_add;
      ep.alpha = epi->alpha;
      for (int t = 0; t < epi->n_add; ++t) {
        SRH_REQUIRE(epi->d_add[t], "spmm_f32: null add[%d]", t);
        ep.add[t] = epi->d_add[t];
        ep.add_scale[t] = epi->add_scale[t];
      }
    }
  }
  if (epi && (epi->flags & SRH_EPI_ADAM)) {
    SRH_REQUIRE(d >= 64 && !(epi->flags & (SRH_EPI_PERTURB | SRH_EPI_MEAN)) && !(epi->scale_flags & SRH_SCALE_OUT) &&
                    !epi->d_row_mark && !epi->noise_d_full,
                "spmm_f32: ADAM runs on whole rows of >= 64 columns, after AXPY only (no PERTURB / MEAN / SCALE_OUT / row marks)");
    SRH_REQUIRE(epi->d_adam_param && epi->d_adam_m && epi->d_adam_v && epi->d_adam_coef, "spmm_f32: ADAM: null table");
    SRH_REQUIRE(epi->d_adam_param != d_x && epi->d_adam_m != d_x && epi->d_adam_v != d_x, "spmm_f32: ADAM updates a table the product reads");
    SRH_REQUIRE(epi->adam_n_clear >= 0 && epi->adam_n_clear <= SRH_MAX_ADAM_CLEAR &&
                    (epi->adam_n_clear == 0 || (epi->d_adam_clear_mark && epi->d_mark_stamp)),
                "spmm_f32: ADAM: at most %d tables to clear, with their row marks and the stamp", SRH_MAX_ADAM_CLEAR);
    SRH_REQUIRE(!epi->d_adam_cursor || (epi->d_mark_stamp != epi->d_adam_cursor && epi->d_mark_stamp != epi->d_adam_cursor + 1),
                "spmm_f32: ADAM: d_mark_stamp must be a copy of the step when the launch advances the cursor");
    ep.adam_p = reinterpret_cast<float4*>(epi->d_adam_param);
    ep.adam_m = reinterpret_cast<float4*>(epi->d_adam_m);
    ep.adam_v = reinterpret_cast<float4*>(epi->d_adam_v);
    ep.adam_coef = epi->d_adam_coef;
    ep.adam_b1 = epi->adam_beta1; ep.adam_b2 = epi->adam_beta2; ep.adam_eps = epi->adam_eps;
    ep.adam_n_clear = epi->adam_n_clear;
    ep.adam_clear_mark = epi->d_adam_clear_mark;
    for (int k = 0; k < epi->adam_n_clear; ++k) {
      SRH_REQUIRE(epi->d_adam_clear[k] && epi->d_adam_clear[k] != d_x, "spmm_f32: ADAM: table %d to clear is null or the product's x", k);
      ep.adam_clear[k] = reinterpret_cast<float4*>(epi->d_adam_clear[k]);
    }
    ep.adam_cursor = epi->d_adam_cursor;
  }
  if (!ep.noise_d_full) { ep.noise_d_full = d; ep.noise_col0 = 0; }
  ep.noise_d_valid = ep.noise_d_full;
  if (epi && epi->noise_d_valid) {
    SRH_REQUIRE(epi->noise_d_valid > 0 && epi->noise_d_valid <= ep.noise_d_full,
                "spmm_f32: noise_d_valid = %d outside (0, %d]", epi->noise_d_valid, ep.noise_d_full);
    SRH_REQUIRE(d >= 64 || epi->noise_d_valid == ep.noise_d_full,
                "spmm_f32: zero-padded rows (noise_d_valid < row width) are served on tables of >= 64 columns, not %d", d);
    ep.noise_d_valid = epi->noise_d_valid;
  }
  SRH_REQUIRE(d > 32 || ep.noise_d_full % 32 == 0 || !(ep.flags & SRH_EPI_PERTURB),
              "spmm_f32: PERTURB on %d-wide rows needs the whole row width (a multiple of 32) in noise_d_full", d);
  return SRH_OK;
}

extern "C" {

srh_status_t srh_spmm_plan_create(srh_spmm_plan_t** out, int64_t n_rows, int64_t n_cols,
                                  const int32_t* h_indptr, int32_t split_len, int64_t xcd_split_row,
                                  const int32_t* h_row_mid) {
  SRH_REQUIRE(out && h_indptr, "spmm_plan_create: null argument");
  SRH_REQUIRE(n_rows > 0 && n_cols > 0, "spmm_plan_create: bad shape");
  SRH_REQUIRE(n_rows < (int64_t(1) << 31) && n_cols < (int64_t(1) << 31), "spmm_plan_create: shape exceeds int32");
  // rows longer than split_len are cut into cooperative segments; 512 sits at the measured optimum
  // (384 / 512 / 768 -> 50.4 / 48.8 / 53.3 us per Yelp2018-shape launch, profiles/r01_i_spmm_short_split_sweep.txt)
  if (split_len <= 0) split_len = 512;
  SRH_REQUIRE(split_len % 64 == 0, "spmm_plan_create: split_len must be a multiple of 64");
  SRH_REQUIRE(h_indptr[0] == 0, "spmm_plan_create: indptr[0] != 0");
  SRH_REQUIRE(xcd_split_row >= 0 && xcd_split_row <= n_rows, "spmm_plan_create: xcd_split_row out of range");
  struct ClsSeg { Seg seg; int8_t half; };
  std::vector<ClsSeg> segs;
  std::vector<Heavy> heavy;
  std::vector<int32_t> slot_owner;
  segs.reserve((size_t)n_rows + 1024);
  int32_t n_slots = 0;
  for (int64_t r = 0; r < n_rows; ++r) {
    const int32_t s = h_indptr[r], e = h_indptr[r + 1];
    if (e < s) { srh::set_error("spmm_plan_create: indptr not monotone at row %lld", (long long)r); return SRH_ERR_INVALID_ARG; }
    // h_row_mid[r] >= 0: the row's entries are ordered [column class 0 | column class 1] and the second
    // part starts at s + h_row_mid[r]; the two parts become separate segments (run on different XCDs).
    // h_row_mid[r] = -1 - c: the row stays whole and goes with column class c.
    const int32_t mid_rel = h_row_mid ? h_row_mid[r] : -1;
    if (mid_rel > e - s) { srh::set_error("spmm_plan_create: row_mid out of range at row %lld", (long long)r); return SRH_ERR_INVALID_ARG; }
    const int8_t whole_half = (mid_rel < 0) ? (int8_t)((-1 - mid_rel) & 1) : 0;
    struct Part { int32_t s, e; int8_t half; };
    Part parts[2];
    int n_parts = 0;
    if (mid_rel < 0) {
      parts[n_parts++] = {s, e, whole_half};
    } else {
      if (mid_rel > 0) parts[n_parts++] = {s, s + mid_rel, 0};
      if (s + mid_rel < e) parts[n_parts++] = {s + mid_rel, e, 1};
      if (n_parts == 0) parts[n_parts++] = {s, e, 0};
    }
    int32_t pieces = 0;
    for (int k = 0; k < n_parts; ++k) pieces += std::max(1, (parts[k].e - parts[k].s + split_len - 1) / split_len);
    if (pieces == 1) {
      segs.push_back({{(int32_t)r, parts[0].s, parts[0].e, -1}, parts[0].half});
      continue;
    }
    heavy.push_back({(int32_t)r, n_slots, pieces, 0});
    for (int k = 0; k < n_parts; ++k) {
      const int32_t np_k = std::max(1, (parts[k].e - parts[k].s + split_len - 1) / split_len);
      for (int32_t q = 0; q < np_k; ++q) {
        segs.push_back({{(int32_t)r, parts[k].s + q * split_len, std::min(parts[k].e, parts[k].s + (q + 1) * split_len), n_slots++},
                        parts[k].half});
        slot_owner.push_back((int32_t)heavy.size() - 1);
      }
    }
  }

  srh_spmm_plan* p = new (std::nothrow) srh_spmm_plan();
  if (!p) { srh::set_error("spmm_plan_create: out of memory"); return SRH_ERR_NOMEM; }
  p->n_rows = n_rows; p->n_cols = n_cols; p->nnz = h_indptr[n_rows];
  p->n_heavy = (int32_t)heavy.size(); p->n_slots = n_slots;
  p->split_len = split_len;
  // ---- coop tasks (long rows / split pieces) then G short rows per task ----
  // tsegs = coop(class 0) ++ coop(class 1) .. ++ short(class 0) ++ short(class 1) .., each longest first (ties keep
  // row order, so neighbouring waves touch neighbouring y rows); one task list per row-group count.
  // Task classes: row class (user rows / item rows of a bipartite adjacency, split at xcd_split_row) x column class.
  // Workgroup b = 4 consecutive tasks runs on XCD b % 8 (observed dispatch rule, performance only); the 8 XCDs are
  // dealt to the classes in equal groups, so each L2 caches only the x rows of one row class AND one column class.
  std::vector<Seg> tsegs;
  std::vector<Task> tasks[5];
  {
    const int n_col_classes = h_row_mid ? 2 : 1;
    const bool two_row = xcd_split_row > 0 && xcd_split_row < n_rows;
    const int NC = (two_row ? 2 : 1) * n_col_classes;             // 1, 2 or 4
    std::vector<Seg> coop[4], shorts[4];
    for (const ClsSeg& sc : segs) {
      const Seg& sgm = sc.seg;
      const int cls = ((two_row && sgm.row >= xcd_split_row) ? n_col_classes : 0) + (n_col_classes > 1 ? sc.half : 0);
      (((sgm.end - sgm.start) > kShortRow || sgm.slot >= 0) ? coop : shorts)[cls].push_back(sgm);
    }
    auto longer = [](const Seg& a, const Seg& b) { return (a.end - a.start) > (b.end - b.start); };
    int32_t coop_off[4], short_off[4];
    for (int c = 0; c < NC; ++c) std::stable_sort(coop[c].begin(), coop[c].end(), longer);
    for (int c = 0; c < NC; ++c) std::stable_sort(shorts[c].begin(), shorts[c].end(), longer);
    for (int c = 0; c < NC; ++c) { coop_off[c] = (int32_t)tsegs.size(); tsegs.insert(tsegs.end(), coop[c].begin(), coop[c].end()); }
    for (int c = 0; c < NC; ++c) { short_off[c] = (int32_t)tsegs.size(); tsegs.insert(tsegs.end(), shorts[c].begin(), shorts[c].end()); }
    for (int gi = 0; gi < 5; ++gi) {
      const int Gr = (gi == 4) ? 16 : (8 >> gi);         // rows per wave for LPR = 8, 16, 32, 64; then LPR = 4
      std::vector<Task>& out_t = tasks[gi];
      size_t blk = 0;
      auto emit = [&](const std::vector<Seg>* lists, int kind, const int32_t* base, int unit) {
        size_t pos[4] = {0, 0, 0, 0};
        auto left = [&](int c) { return lists[c].size() - pos[c]; };
        for (;;) {
          size_t total = 0;
          for (int c = 0; c < NC; ++c) total += left(c);
          if (total == 0) break;
          int want = (int)((blk % 8) / (8 / NC));        // the class this workgroup's XCD belongs to
          for (int w = 0; w < 4; ++w) {
            int c = want;
            if (left(c) == 0) {                          // its list ran dry: help the fullest one
              for (int k = 0; k < NC; ++k) if (left(k) > left(c)) c = k;
              if (left(c) == 0) break;
            }
            const int32_t cnt = (int32_t)std::min<size_t>(unit, left(c));
            out_t.push_back({kind, base[c] + (int32_t)pos[c], cnt, 0});
            pos[c] += cnt;
          }
          ++blk;
        }
      };
      emit(coop, 0, coop_off, 1);
      emit(shorts, 1, short_off, Gr);
    }
  }
  for (int k = 0; k < 16; ++k) tsegs.push_back({0, 0, 0, -1});       // padding records (see d_tsegs)
  hipError_t err = hipMalloc(&p->d_tsegs, sizeof(Seg) * tsegs.size());
  if (err == hipSuccess) err = hipMemcpy(p->d_tsegs, tsegs.data(), sizeof(Seg) * tsegs.size(), hipMemcpyHostToDevice);
  for (int gi = 0; gi < 5 && err == hipSuccess; ++gi) {
    p->n_tasks[gi] = (int32_t)tasks[gi].size();
    err = hipMalloc(&p->d_tasks[gi], sizeof(Task) * std::max<size_t>(1, tasks[gi].size()));
    if (err == hipSuccess && !tasks[gi].empty())
      err = hipMemcpy(p->d_tasks[gi], tasks[gi].data(), sizeof(Task) * tasks[gi].size(), hipMemcpyHostToDevice);
  }
  // spmm_rows_kernel (d = 64 / 128 / 256) reads one self-contained 64-byte record per wave
  for (int gi = 1; gi <= 3 && err == hipSuccess; ++gi) {
    const int Gr = 8 >> gi;
    std::vector<Task64> t64(tasks[gi].size());
    for (size_t k = 0; k < tasks[gi].size(); ++k) {
      const Task& tk = tasks[gi][k];
      Task64 r{};
      r.kind = tk.kind; r.count = tk.count; r.slot = -1;
      for (int q = 0; q < 4; ++q) {
        const bool have = tk.kind == 1 ? (q < tk.count && q < Gr) : q == 0;
        const Seg& sg = tsegs[tk.first + (have ? q : 0)];
        r.row[q] = sg.row; r.start[q] = sg.start; r.end[q] = have ? sg.end : sg.start;
        if (q == 0) r.slot = sg.slot;
      }
      t64[k] = r;
    }
    err = hipMalloc(&p->d_tasks64[gi], sizeof(Task64) * std::max<size_t>(1, t64.size()));
    if (err == hipSuccess && !t64.empty())
      err = hipMemcpy(p->d_tasks64[gi], t64.data(), sizeof(Task64) * t64.size(), hipMemcpyHostToDevice);
    p->n_run[gi] = (int32_t)t64.size();
    p->cap64[gi] = std::max<size_t>(1, t64.size());
    p->h_tasks64[gi] = std::move(t64);
  }
  if (err == hipSuccess && !heavy.empty()) {
    err = hipMalloc(&p->d_heavy, sizeof(Heavy) * heavy.size());
    if (err == hipSuccess) err = hipMemcpy(p->d_heavy, heavy.data(), sizeof(Heavy) * heavy.size(), hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMalloc(&p->d_partial, sizeof(float) * 256 * (size_t)n_slots);
    if (err == hipSuccess) err = hipMalloc(&p->d_slot_owner, sizeof(int32_t) * slot_owner.size());
    if (err == hipSuccess) err = hipMemcpy(p->d_slot_owner, slot_owner.data(), sizeof(int32_t) * slot_owner.size(), hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMalloc(&p->d_tickets, sizeof(int32_t) * heavy.size());
    if (err == hipSuccess) err = hipMemset(p->d_tickets, 0, sizeof(int32_t) * heavy.size());
  }
  if (err != hipSuccess) {
    srh::set_error("spmm_plan_create: %s", hipGetErrorString(err));
    srh_spmm_plan_destroy(p);
    return SRH_ERR_HIP;
  }
  *out = p;
  return SRH_OK;
}

void srh_spmm_plan_destroy(srh_spmm_plan_t* p) {
  if (!p) return;
  if (p->d_heavy) (void)hipFree(p->d_heavy);
  if (p->d_partial) (void)hipFree(p->d_partial);
  if (p->d_slot_owner) (void)hipFree(p->d_slot_owner);
  if (p->d_tickets) (void)hipFree(p->d_tickets);
  if (p->d_tsegs) (void)hipFree(p->d_tsegs);
  for (int gi = 0; gi < 5; ++gi) if (p->d_tasks[gi]) (void)hipFree(p->d_tasks[gi]);
  for (int gi = 0; gi < 5; ++gi) if (p->d_tasks64[gi]) (void)hipFree(p->d_tasks64[gi]);
  delete p;
}

srh_status_t srh_spmm3_f32(const srh_spmm_plan_t* plan, const int32_t* d_indices, const float* d_vals0,
                           const float* d_vals1, const float* d_vals2, const float* d_x, float* d_y0,
                           float* d_y1, float* d_y2, int32_t d, void* stream) {
  SRH_REQUIRE(plan && d_indices && d_vals0 && d_vals1 && d_vals2 && d_x && d_y0 && d_y1 && d_y2, "spmm3_f32: null argument");
  SRH_REQUIRE(d_x != d_y0 && d_x != d_y1 && d_x != d_y2 && d_y0 != d_y1 && d_y0 != d_y2 && d_y1 != d_y2,
              "spmm3_f32: x and the three outputs must be distinct");
  if (d != 64) {
    srh::set_error("spmm3_f32: d=%d unsupported (d = 64 only)", d);
    return SRH_ERR_UNSUPPORTED;
  }
  constexpr int gi = 1;                      // LPR = 16
  spmm_rows3_kernel<16><<<(plan->n_tasks[gi] + 3) / 4, 256, 0, srh::as_stream(stream)>>>(
      plan->d_tasks[gi], plan->n_tasks[gi], plan->d_tsegs, d_indices, d_vals0, d_vals1, d_vals2,
      reinterpret_cast<const float4*>(d_x), reinterpret_cast<float4*>(d_y0), reinterpret_cast<float4*>(d_y1),
      reinterpret_cast<float4*>(d_y2), reinterpret_cast<float4*>(plan->d_partial), plan->d_heavy, plan->d_slot_owner,
      plan->d_tickets);
  SRH_LAUNCH_CHECK();
  return SRH_OK;
}

srh_status_t srh_spmm_f32(const srh_spmm_plan_t* plan, const int32_t* d_indptr,
                          const int32_t* d_indices, const float* d_vals, const float* d_x,
                          float* d_y, int32_t d, const srh_spmm_epilogue_t* epi, void* stream) {
  return srh_spmm_f32_with_fetch(plan, d_indptr, d_indices, d_vals, d_x, d_y, d, epi, nullptr, stream);
}

static srh_status_t spmm_launch(const srh_spmm_plan_t* plan, const int32_t* d_indices, const float* d_vals, const float* d_x,
                                float* d_y, int32_t d, const srh_spmm_epilogue_t* epi, const srh_batch_fetch_args_t* fetch,
                                unsigned long long* d_stamps, void* stream, bool floor_only = false);

srh_status_t srh_spmm_f32_with_fetch(const srh_spmm_plan_t* plan, const int32_t* d_indptr, const int32_t* d_indices,
                                     const float* d_vals, const float* d_x, float* d_y, int32_t d,
                                     const srh_spmm_epilogue_t* epi, const srh_batch_fetch_args_t* fetch, void* stream) {
  (void)d_indptr;  // the schedule in `plan` already encodes the row extents
  return spmm_launch(plan, d_indices, d_vals, d_x, d_y, d, epi, fetch, nullptr, stream);
}

srh_status_t srh_spmm_f32_probe(const srh_spmm_plan_t* plan, const int32_t* d_indices, const float* d_vals, const float* d_x,
                                float* d_y, int32_t d, const srh_spmm_epilogue_t* epi, uint64_t* d_stamps, void* stream) {
  SRH_REQUIRE(d_stamps, "spmm_f32_probe: null stamp buffer");
  SRH_REQUIRE(d == 64 || d == 128 || d == 256, "spmm_f32_probe: d=%d unsupported (64, 128 or 256)", d);
  return spmm_launch(plan, d_indices, d_vals, d_x, d_y, d, epi, nullptr, reinterpret_cast<unsigned long long*>(d_stamps), stream);
}

srh_status_t srh_spmm_gather_bound(const srh_spmm_plan_t* plan, const int32_t* d_indices, const float* d_x, float* d_scratch,
                                   int32_t d, void* stream) {
  SRH_REQUIRE(d == 64 || d == 128 || d == 256, "spmm_gather_bound: d=%d unsupported (64, 128 or 256)", d);
  SRH_REQUIRE(d_scratch, "spmm_gather_bound: null scratch row");
  return spmm_launch(plan, d_indices, nullptr, d_x, d_scratch, d, nullptr, nullptr, nullptr, stream, true);
}

srh_status_t srh_gather_floor_probe(const int32_t* d_indices, int64_t n_idx, const float* d_x, int64_t n_x_rows, int32_t d,
                                    int32_t blocks, float* d_sink, void* stream) {
  SRH_REQUIRE(d_indices && d_x && d_sink && n_idx > 0 && n_x_rows > 0, "gather_floor_probe: null / empty argument");
  SRH_REQUIRE(d == 64 || d == 128 || d == 256, "gather_floor_probe: d=%d unsupported (64, 128 or 256)", d);
  SRH_REQUIRE(n_x_rows * (int64_t)d * 4 < (int64_t(1) << 32), "gather_floor_probe: x must be smaller than 4 GiB");
  SRH_REQUIRE(blocks > 0 && blocks <= (1 << 20), "gather_floor_probe: bad grid");
  hipStream_t st = srh::as_stream(stream);
  const float4* X = reinterpret_cast<const float4*>(d_x);
  float4* sink = reinterpret_cast<float4*>(d_sink);
  if (d == 64) gather_floor_kernel<16><<<blocks, 256, 0, st>>>(X, d_indices, (long)n_idx, sink);
  else if (d == 128) gather_floor_kernel<32><<<blocks, 256, 0, st>>>(X, d_indices, (long)n_idx, sink);
  else gather_floor_kernel<64><<<blocks, 256, 0, st>>>(X, d_indices, (long)n_idx, sink);
  SRH_LAUNCH_CHECK();
  return SRH_OK;
}

int32_t srh_spmm_plan_run_tasks(const srh_spmm_plan_t* plan, int32_t d) {
  if (!plan) return -1;
  return d == 64 ? plan->n_run[1] : d == 128 ? plan->n_run[2] : d == 256 ? plan->n_run[3] : -1;
}

srh_status_t srh_spmm_plan_set_xcd_shares(srh_spmm_plan_t* plan, int32_t d, const int32_t* h_blocks_per_xcd) {
  SRH_REQUIRE(plan, "spmm_plan_set_xcd_shares: null plan");
  SRH_REQUIRE(d == 64 || d == 128 || d == 256, "spmm_plan_set_xcd_shares: d=%d unsupported (64, 128 or 256)", d);
  const int gi = d == 64 ? 1 : d == 128 ? 2 : 3;
  const std::vector<Task64>& canon = plan->h_tasks64[gi];
  const size_t nb = (canon.size() + 3) / 4;                 // canonical blocks (4 records each)
  Task64 empty{};
  empty.kind = 1; empty.count = 0; empty.slot = -1;
  auto block = [&](size_t b, int w) -> const Task64& { return b * 4 + w < canon.size() ? canon[b * 4 + w] : empty; };
  std::vector<Task64> list;
  if (!h_blocks_per_xcd) {
    list = canon;                                            // back to the canonical list
  } else {
    int64_t total = 0;
    for (int k = 0; k < 8; ++k) {
      SRH_REQUIRE(h_blocks_per_xcd[k] >= 0, "spmm_plan_set_xcd_shares: negative share");
      total += h_blocks_per_xcd[k];
    }
    SRH_REQUIRE((size_t)total == nb, "spmm_plan_set_xcd_shares: shares add up to %lld blocks, the plan has %lld",
                (long long)total, (long long)nb);
    // canonical queues: block b runs on XCD b % 8.  Queues above their share give up their LAST blocks (the short rows of
    // the tail: locality matters least there), queues below theirs append them in that order.
    std::vector<size_t> q[8], pool;
    for (size_t b = 0; b < nb; ++b) q[b % 8].push_back(b);
    for (int k = 0; k < 8; ++k)
      while ((int64_t)q[k].size() > h_blocks_per_xcd[k]) { pool.push_back(q[k].back()); q[k].pop_back(); }
    std::reverse(pool.begin(), pool.end());
    for (int k = 0; k < 8; ++k)
      while ((int64_t)q[k].size() < h_blocks_per_xcd[k]) { q[k].push_back(pool.back()); pool.pop_back(); }
    size_t depth = 0;
    for (int k = 0; k < 8; ++k) depth = std::max(depth, q[k].size());
    list.assign(depth * 8 * 4, empty);
    for (int k = 0; k < 8; ++k)
      for (size_t pos = 0; pos < q[k].size(); ++pos)
        for (int w = 0; w < 4; ++w) list[(pos * 8 + k) * 4 + w] = block(q[k][pos], w);
  }
  if (list.size() > plan->cap64[gi]) {                       // (hipFree synchronises: never inside a stream capture)
    Task64* fresh = nullptr;
    hipError_t err = hipMalloc(&fresh, sizeof(Task64) * list.size());
    if (err != hipSuccess) { srh::set_error("spmm_plan_set_xcd_shares: %s", hipGetErrorString(err)); return SRH_ERR_HIP; }
    (void)hipFree(plan->d_tasks64[gi]);
    plan->d_tasks64[gi] = fresh;
    plan->cap64[gi] = list.size();
  }
  hipError_t err = hipDeviceSynchronize();                   // no launch may still be reading the old list
  if (err == hipSuccess && !list.empty())
    err = hipMemcpy(plan->d_tasks64[gi], list.data(), sizeof(Task64) * list.size(), hipMemcpyHostToDevice);
  if (err != hipSuccess) { srh::set_error("spmm_plan_set_xcd_shares: %s", hipGetErrorString(err)); return SRH_ERR_HIP; }
  plan->n_run[gi] = (int32_t)list.size();
  return SRH_OK;
}

static srh_status_t spmm_launch(const srh_spmm_plan_t* plan, const int32_t* d_indices, const float* d_vals, const float* d_x,
                                float* d_y, int32_t d, const srh_spmm_epilogue_t* epi, const srh_batch_fetch_args_t* fetch,
                                unsigned long long* d_stamps, void* stream, bool floor_only) {
  srh_batch_fetch_args_t fetch_args{};
  int n_fetch = 0;
  if (fetch) {
    SRH_REQUIRE(d >= 64 && !(epi && (epi->d_col_mark || epi->d_row_mark)),
                "spmm_f32_with_fetch: d >= 64, and the product must not depend on the batch's marks");
    if (srh_status_t rc = srh::check_fetch_args(fetch, fetch_args)) return rc;
    n_fetch = srh::kFetchBlocks;
  }
  SRH_REQUIRE(plan && d_indices && d_x && d_y, "spmm_f32: null argument");
  SRH_REQUIRE(d_vals || (d >= 64 && !(epi && epi->d_col_mark)),
              "spmm_f32: a pattern matrix (d_vals == NULL) needs d >= 64 and no column marks");
  SRH_REQUIRE(srh::dim_supported(d) || d == 8 || d == 16, "spmm_f32: d=%d unsupported (need 8, 16, 32, 64, 128 or 256)", d);
  SRH_REQUIRE(d_x != d_y, "spmm_f32: x and y must not alias");
  SRH_REQUIRE(plan->n_cols * (int64_t)d * 4 < (int64_t(1) << 32), "spmm_f32: x (%lld rows x %d) must be smaller than 4 GiB",
              (long long)plan->n_cols, d);
  DevEpilogue ep{};
  if (srh_status_t rc = translate_epilogue(epi, d, d_x, d_y, ep)) return rc;
  hipStream_t st = srh::as_stream(stream);
  // one kernel per table width: a row-group of d/4 lanes per gathered x row (d >= 16), two lanes per row at d = 8
#define SRH_LAUNCH(KERNEL, GI, XT, YT)                                                                              \
  KERNEL<<<(plan->n_tasks[GI] + 3) / 4, 256, 0, st>>>(plan->d_tasks[GI], plan->n_tasks[GI], plan->d_tsegs, d_indices,  \
                                                      d_vals, reinterpret_cast<const XT*>(d_x), reinterpret_cast<YT*>(d_y), \
                                                      reinterpret_cast<YT*>(plan->d_partial), plan->d_heavy,          \
                                                      plan->d_slot_owner, plan->d_tickets, ep)
  switch (d) {
    case 8: SRH_LAUNCH(spmm_pair_kernel, 0, float, float); break;
    case 16: SRH_LAUNCH(spmm_slice_kernel<4>, 4, float4, float4); break;
    case 32: SRH_LAUNCH(spmm_slice_kernel<8>, 0, float4, float4); break;
    default: {
      // (the gather offsets carry "no gather" in their sign bit: the table must stay below 2 GiB)
      SRH_REQUIRE(plan->n_cols * (int64_t)d * 4 < (int64_t(1) << 31), "spmm_f32: x (%lld rows x %d) must be smaller than 2 GiB",
                  (long long)plan->n_cols, d);
#define SRH_LAUNCH_ROWS(LPRV, GI, CM, PR, ...)                                                                       \
  spmm_rows_kernel<LPRV, CM, PR, ##__VA_ARGS__><<<(plan->n_run[GI] + 3) / 4 + n_fetch, 256, 0, st>>>(           \
      plan->d_tasks64[GI], plan->n_run[GI], d_indices, d_vals, reinterpret_cast<const float4*>(d_x),                \
      reinterpret_cast<float4*>(d_y), reinterpret_cast<float4*>(plan->d_partial), plan->d_heavy, plan->d_slot_owner, \
      plan->d_tickets, ep, n_fetch, fetch_args)
      ep.stamps = d_stamps;
      SRH_REQUIRE(!(ep.flags & SRH_EPI_ADAM) || !(floor_only || d_stamps || ep.col_mark),
                  "spmm_f32: ADAM runs on the plain launch (no column marks, not under the probes)");
      if (floor_only) {
        if (d == 64) SRH_LAUNCH_ROWS(16, 1, false, false, false, true);
        else if (d == 128) SRH_LAUNCH_ROWS(32, 2, false, false, false, true);
        else SRH_LAUNCH_ROWS(64, 3, false, false, false, true);
      } else if (d_stamps) {
        SRH_REQUIRE(!ep.col_mark, "spmm_f32_probe: no column marks");
        if (d == 64) SRH_LAUNCH_ROWS(16, 1, false, true);
        else if (d == 128) SRH_LAUNCH_ROWS(32, 2, false, true);
        else SRH_LAUNCH_ROWS(64, 3, false, true);
      } else if (ep.col_mark) {
        if (d == 64) SRH_LAUNCH_ROWS(16, 1, true, false);
        else if (d == 128) SRH_LAUNCH_ROWS(32, 2, true, false);
        else SRH_LAUNCH_ROWS(64, 3, true, false);
      } else if (ep.flags & SRH_EPI_ADAM) {
        if (d == 64) SRH_LAUNCH_ROWS(16, 1, false, false, false, false, true);
        else if (d == 128) SRH_LAUNCH_ROWS(32, 2, false, false, false, false, true);
        else SRH_LAUNCH_ROWS(64, 3, false, false, false, false, true);
      } else if (ep.row_mark && d == 64) {
        SRH_LAUNCH_ROWS(16, 1, false, false, true);              // row-masked launch: late prefetch (measured at d = 64)
      } else {
        if (d == 64) SRH_LAUNCH_ROWS(16, 1, false, false);
        else if (d == 128) SRH_LAUNCH_ROWS(32, 2, false, false);
        else SRH_LAUNCH_ROWS(64, 3, false, false);
      }
#undef SRH_LAUNCH_ROWS
    }
  }
#undef SRH_LAUNCH
  SRH_LAUNCH_CHECK();
  return SRH_OK;
}

}  // extern "C"
