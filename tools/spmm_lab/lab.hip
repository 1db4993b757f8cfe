// SpMM laboratory (tools only -- NOT part of libselfrec_hip.so): candidate kernels for the d = 64 propagation
// launch, built next to the product kernels (this file #includes csrc/spmm.hip, so it shares DevEpilogue,
// row_epilogue, the plan structure and the split-row hand-off) and driven by tools/spmm_lab/run.py, which
// checks every variant against the product launch and times it.  A variant that wins is ported into
// csrc/spmm.hip; the measurements go to profiles/ and DESIGN.md.  VERDICT r01 "Next round" #4.
//
// Variants (lab_spmm `variant` argument):
//   1  asm inner loop.  The compiler's gather loop spends ~12 VALU + 4 SALU instructions per gathered x row
//      (zero-init movs for the DPP 'old' operand, a v_lshl_or per address, 4 zero-fill movs + saveexec/branch
//      per predicated load).  Here: address = v_or_b32_dpp(col << 8, lane offset) -- the DPP broadcast and the
//      address arithmetic are ONE instruction --, value = one v_mov_b32_dpp, predicate = v_cmpx on the value
//      (padding / dropped / dead entries issue no gather; their destination registers keep older, finite x
//      values and are multiplied by 0), loads counted with explicit vmcnt waits so FMAs start as rows arrive.
//   2  variant 1 + one 64-byte task record per wave read with scalar loads (the product reads a 16-byte task,
//      then 16-byte segment records with vector loads: two dependent vector-memory round trips before the first
//      (col, val) load; here one s_load_dwordx16), next chunk's (col, val) in flight under the current gathers.
//   3  variant 2 without values (timing probe for the value-free form A_ij = d_i^-1/2 d_j^-1/2: x pre-scaled,
//      y scaled per row): entries carry only a column, padding = sign bit; compare against the product launch
//      on an all-ones value array.
//      (x must carry one extra all-zero row: padding entries point at it and every gather is unconditional)
//   6  variant 2 with unconditional gathers (padding -> the zero row, value 0): what the predicate costs / saves.
//  20 / 21 / 22  knock-outs of variant 2: no epilogue on short rows / no x gathers / no (col, val) loads (hashed columns)
//   4  variant 2 with the column-activity test on a BITMAP (1 bit per column, L1-resident 8.7 KB at the Yelp2018
//      shape) instead of one 4-byte mark gather per entry;  5  the same with the bitmap staged in LDS.
#include "../../selfrec_amd/csrc/spmm.hip"

namespace lab {

using namespace srh;

// (Task64: the product's record, csrc/spmm.hip)

typedef float floatx4_t __attribute__((ext_vector_type(4)));

// MODE 60 (timing probe of MODE 2): per wave {begin, end} on the chip-wide 100 MHz clock and the XCD it ran on
__device__ unsigned long long* g_probe = nullptr;

#define LAB_DPP_OR(T)                                                                                          \
  asm volatile("v_or_b32_dpp %0, %1, %2 row_newbcast:" #T " row_mask:0xf bank_mask:0xf" : "=v"(off[T & 7]) : "v"(cs), "v"(sub16))
#define LAB_DPP_MOV(T)                                                                                         \
  asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:" #T " row_mask:0xf bank_mask:0xf" : "=v"(vv[T & 7]) : "v"(v))

// Eight x rows at byte offsets off[0..7] of X; an entry whose offset has the sign bit set (padding, dropped edge,
// dead column: its value is 0) issues no gather -- its destination keeps an older, finite x row that is then
// multiplied by 0.  One exec save per eight loads; v_cmpx writes exec directly (no saveexec / branch / zero-fill).
__device__ __forceinline__ void pred_load8(floatx4_t& x0, floatx4_t& x1, floatx4_t& x2, floatx4_t& x3, floatx4_t& x4,
                                           floatx4_t& x5, floatx4_t& x6, floatx4_t& x7, const unsigned (&off)[8], const void* X) {
  unsigned long long save;
  asm volatile(
      "s_mov_b64 %[sv], exec\n\t"
      "v_cmpx_le_i32_e32 0, %[o0]\n\tglobal_load_dwordx4 %[x0], %[o0], %[b]\n\ts_mov_b64 exec, %[sv]\n\t"
      "v_cmpx_le_i32_e32 0, %[o1]\n\tglobal_load_dwordx4 %[x1], %[o1], %[b]\n\ts_mov_b64 exec, %[sv]\n\t"
      "v_cmpx_le_i32_e32 0, %[o2]\n\tglobal_load_dwordx4 %[x2], %[o2], %[b]\n\ts_mov_b64 exec, %[sv]\n\t"
      "v_cmpx_le_i32_e32 0, %[o3]\n\tglobal_load_dwordx4 %[x3], %[o3], %[b]\n\ts_mov_b64 exec, %[sv]\n\t"
      "v_cmpx_le_i32_e32 0, %[o4]\n\tglobal_load_dwordx4 %[x4], %[o4], %[b]\n\ts_mov_b64 exec, %[sv]\n\t"
      "v_cmpx_le_i32_e32 0, %[o5]\n\tglobal_load_dwordx4 %[x5], %[o5], %[b]\n\ts_mov_b64 exec, %[sv]\n\t"
      "v_cmpx_le_i32_e32 0, %[o6]\n\tglobal_load_dwordx4 %[x6], %[o6], %[b]\n\ts_mov_b64 exec, %[sv]\n\t"
      "v_cmpx_le_i32_e32 0, %[o7]\n\tglobal_load_dwordx4 %[x7], %[o7], %[b]\n\ts_mov_b64 exec, %[sv]\n\t"
      "s_nop 4"        // VALU-written exec -> DPP op needs 5 wait states (the value broadcasts follow)
      : [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2), [x3] "+v"(x3), [x4] "+v"(x4), [x5] "+v"(x5),
        [x6] "+v"(x6), [x7] "+v"(x7), [sv] "=&s"(save)
      : [o0] "v"(off[0]), [o1] "v"(off[1]), [o2] "v"(off[2]), [o3] "v"(off[3]), [o4] "v"(off[4]), [o5] "v"(off[5]),
        [o6] "v"(off[6]), [o7] "v"(off[7]), [b] "s"(X)
      : "memory", "vcc");
}
// unconditional form (padding entries point at an all-zero x row appended to the table)
__device__ __forceinline__ void plain_load(floatx4_t& xr, unsigned off, const void* X) {
  asm volatile("global_load_dwordx4 %[x], %[off], %[base]" : [x] "=v"(xr) : [off] "v"(off), [base] "s"(X) : "memory");
}

#define LAB_WAIT(N, XR) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(XR))

typedef float floatx2_t __attribute__((ext_vector_type(2)));
struct Acc { floatx2_t lo, hi; };       // columns 0-1 / 2-3 of this lane's float4 of the output row

// acc += value * x for one gathered row, as soon as it has landed (N loads may still be in flight): two packed FMAs.
// Two consecutive entries keep their values in ONE even-aligned register pair (SEL = 0: low dword, 1: high dword,
// via op_sel) -- a pair per entry, as the compiler allocates them, costs 8 extra VGPRs and a wave per SIMD.
#define LAB_FMA(N, SEL, VP, XR)                                                                                 \
  asm volatile("s_waitcnt vmcnt(" #N ")\n\t"                                                                    \
               "v_pk_fma_f32 %[lo], %[vp], %[xlo], %[lo] op_sel:[" #SEL ",0,0] op_sel_hi:[" #SEL ",1,1]\n\t"      \
               "v_pk_fma_f32 %[hi], %[vp], %[xhi], %[hi] op_sel:[" #SEL ",0,0] op_sel_hi:[" #SEL ",1,1]"          \
               : [lo] "+v"(acc.lo), [hi] "+v"(acc.hi)                                                           \
               : [vp] "v"(VP), [xlo] "v"(__builtin_shufflevector(XR, XR, 0, 1)), [xhi] "v"(__builtin_shufflevector(XR, XR, 2, 3)))

// entries T0 .. T0+7 of the 16 this lane's DPP row holds: cs = col << 8 (bytes of a 256-byte x row; sign bit = no
// gather), v = value.  xx[] persists across calls (stale rows are finite).  HI selects row_newbcast 8..15.
// The value broadcasts are issued AFTER the loads: VALU work under the memory latency, and off[] / vv[] never live
// together.
template <bool HI, bool PRED, bool NOLOAD = false>
__device__ __forceinline__ void gather8_asm(unsigned cs, float v, unsigned sub16, const void* X, floatx4_t (&xx)[8],
                                            Acc& acc) {
  unsigned off[8];
  float vv[8];
  // (VALU write -> DPP read of the same VGPR needs 2 wait states; inline asm is invisible to the hazard recogniser)
  asm volatile("s_nop 1" : "+v"(cs), "+v"(v));
  if (!HI) {
    LAB_DPP_OR(0); LAB_DPP_OR(1); LAB_DPP_OR(2); LAB_DPP_OR(3); LAB_DPP_OR(4); LAB_DPP_OR(5); LAB_DPP_OR(6); LAB_DPP_OR(7);
  } else {
    LAB_DPP_OR(8); LAB_DPP_OR(9); LAB_DPP_OR(10); LAB_DPP_OR(11); LAB_DPP_OR(12); LAB_DPP_OR(13); LAB_DPP_OR(14); LAB_DPP_OR(15);
  }
  if (NOLOAD) {
    asm volatile("" :: "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "v"(off[4]), "v"(off[5]), "v"(off[6]), "v"(off[7]));
  } else if (PRED) {
    pred_load8(xx[0], xx[1], xx[2], xx[3], xx[4], xx[5], xx[6], xx[7], off, X);
  } else {
#pragma unroll
    for (int t = 0; t < 8; ++t) plain_load(xx[t], off[t], X);
  }
  if (!HI) {
    LAB_DPP_MOV(0); LAB_DPP_MOV(1); LAB_DPP_MOV(2); LAB_DPP_MOV(3); LAB_DPP_MOV(4); LAB_DPP_MOV(5); LAB_DPP_MOV(6); LAB_DPP_MOV(7);
  } else {
    LAB_DPP_MOV(8); LAB_DPP_MOV(9); LAB_DPP_MOV(10); LAB_DPP_MOV(11); LAB_DPP_MOV(12); LAB_DPP_MOV(13); LAB_DPP_MOV(14); LAB_DPP_MOV(15);
  }
  const floatx2_t p0 = {vv[0], vv[1]}, p1 = {vv[2], vv[3]}, p2 = {vv[4], vv[5]}, p3 = {vv[6], vv[7]};
  LAB_FMA(7, 0, p0, xx[0]);
  LAB_FMA(6, 1, p0, xx[1]);
  LAB_FMA(5, 0, p1, xx[2]);
  LAB_FMA(4, 1, p1, xx[3]);
  LAB_FMA(3, 0, p2, xx[4]);
  LAB_FMA(2, 1, p2, xx[5]);
  LAB_FMA(1, 0, p3, xx[6]);
  LAB_FMA(0, 1, p3, xx[7]);
}

// all 16 entries of the DPP row in flight at once (xx[16]): twice the bytes in flight per wave, ~100 VGPRs
__device__ __forceinline__ void gather16_asm(unsigned cs, float v, unsigned sub16, const void* X, floatx4_t (&xx)[16], Acc& acc) {
  asm volatile("s_nop 1" : "+v"(cs), "+v"(v));
  {
    unsigned off[8];
    LAB_DPP_OR(0); LAB_DPP_OR(1); LAB_DPP_OR(2); LAB_DPP_OR(3); LAB_DPP_OR(4); LAB_DPP_OR(5); LAB_DPP_OR(6); LAB_DPP_OR(7);
    pred_load8(xx[0], xx[1], xx[2], xx[3], xx[4], xx[5], xx[6], xx[7], off, X);
  }
  {
    unsigned off[8];
    LAB_DPP_OR(8); LAB_DPP_OR(9); LAB_DPP_OR(10); LAB_DPP_OR(11); LAB_DPP_OR(12); LAB_DPP_OR(13); LAB_DPP_OR(14); LAB_DPP_OR(15);
    pred_load8(xx[8], xx[9], xx[10], xx[11], xx[12], xx[13], xx[14], xx[15], off, X);
  }
  {
    float vv[8];
    LAB_DPP_MOV(0); LAB_DPP_MOV(1); LAB_DPP_MOV(2); LAB_DPP_MOV(3); LAB_DPP_MOV(4); LAB_DPP_MOV(5); LAB_DPP_MOV(6); LAB_DPP_MOV(7);
    const floatx2_t p0 = {vv[0], vv[1]}, p1 = {vv[2], vv[3]}, p2 = {vv[4], vv[5]}, p3 = {vv[6], vv[7]};
    LAB_FMA(15, 0, p0, xx[0]); LAB_FMA(14, 1, p0, xx[1]); LAB_FMA(13, 0, p1, xx[2]); LAB_FMA(12, 1, p1, xx[3]);
    LAB_FMA(11, 0, p2, xx[4]); LAB_FMA(10, 1, p2, xx[5]); LAB_FMA(9, 0, p3, xx[6]); LAB_FMA(8, 1, p3, xx[7]);
  }
  {
    float vv[8];
    LAB_DPP_MOV(8); LAB_DPP_MOV(9); LAB_DPP_MOV(10); LAB_DPP_MOV(11); LAB_DPP_MOV(12); LAB_DPP_MOV(13); LAB_DPP_MOV(14); LAB_DPP_MOV(15);
    const floatx2_t p0 = {vv[0], vv[1]}, p1 = {vv[2], vv[3]}, p2 = {vv[4], vv[5]}, p3 = {vv[6], vv[7]};
    LAB_FMA(7, 0, p0, xx[8]); LAB_FMA(6, 1, p0, xx[9]); LAB_FMA(5, 0, p1, xx[10]); LAB_FMA(4, 1, p1, xx[11]);
    LAB_FMA(3, 0, p2, xx[12]); LAB_FMA(2, 1, p2, xx[13]); LAB_FMA(1, 0, p3, xx[14]); LAB_FMA(0, 1, p3, xx[15]);
  }
}

// the low 8 entries on a 16-row register array (DEPTH 2 kernels, chunks with <= 8 entries left)
__device__ __forceinline__ void gather8_lo16(unsigned cs, float v, unsigned sub16, const void* X, floatx4_t (&xx)[16], Acc& acc) {
  unsigned off[8];
  float vv[8];
  asm volatile("s_nop 1" : "+v"(cs), "+v"(v));
  LAB_DPP_OR(0); LAB_DPP_OR(1); LAB_DPP_OR(2); LAB_DPP_OR(3); LAB_DPP_OR(4); LAB_DPP_OR(5); LAB_DPP_OR(6); LAB_DPP_OR(7);
  pred_load8(xx[0], xx[1], xx[2], xx[3], xx[4], xx[5], xx[6], xx[7], off, X);
  LAB_DPP_MOV(0); LAB_DPP_MOV(1); LAB_DPP_MOV(2); LAB_DPP_MOV(3); LAB_DPP_MOV(4); LAB_DPP_MOV(5); LAB_DPP_MOV(6); LAB_DPP_MOV(7);
  const floatx2_t p0 = {vv[0], vv[1]}, p1 = {vv[2], vv[3]}, p2 = {vv[4], vv[5]}, p3 = {vv[6], vv[7]};
  LAB_FMA(7, 0, p0, xx[0]); LAB_FMA(6, 1, p0, xx[1]); LAB_FMA(5, 0, p1, xx[2]); LAB_FMA(4, 1, p1, xx[3]);
  LAB_FMA(3, 0, p2, xx[4]); LAB_FMA(2, 1, p2, xx[5]); LAB_FMA(1, 0, p3, xx[6]); LAB_FMA(0, 1, p3, xx[7]);
}

// value-free: every entry (padding included: it points at the zero row) is gathered and added
template <bool HI>
__device__ __forceinline__ void gather8_novals(unsigned cs, unsigned sub16, const void* X, floatx4_t (&xx)[8],
                                               Acc& acc) {
  unsigned off[8];
  asm volatile("s_nop 1" : "+v"(cs));
  if (!HI) {
    LAB_DPP_OR(0); LAB_DPP_OR(1); LAB_DPP_OR(2); LAB_DPP_OR(3); LAB_DPP_OR(4); LAB_DPP_OR(5); LAB_DPP_OR(6); LAB_DPP_OR(7);
  } else {
    LAB_DPP_OR(8); LAB_DPP_OR(9); LAB_DPP_OR(10); LAB_DPP_OR(11); LAB_DPP_OR(12); LAB_DPP_OR(13); LAB_DPP_OR(14); LAB_DPP_OR(15);
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) plain_load(xx[t], off[t], X);
  LAB_WAIT(7, xx[0]); acc.lo += __builtin_shufflevector(xx[0], xx[0], 0, 1); acc.hi += __builtin_shufflevector(xx[0], xx[0], 2, 3);
  LAB_WAIT(6, xx[1]); acc.lo += __builtin_shufflevector(xx[1], xx[1], 0, 1); acc.hi += __builtin_shufflevector(xx[1], xx[1], 2, 3);
  LAB_WAIT(5, xx[2]); acc.lo += __builtin_shufflevector(xx[2], xx[2], 0, 1); acc.hi += __builtin_shufflevector(xx[2], xx[2], 2, 3);
  LAB_WAIT(4, xx[3]); acc.lo += __builtin_shufflevector(xx[3], xx[3], 0, 1); acc.hi += __builtin_shufflevector(xx[3], xx[3], 2, 3);
  LAB_WAIT(3, xx[4]); acc.lo += __builtin_shufflevector(xx[4], xx[4], 0, 1); acc.hi += __builtin_shufflevector(xx[4], xx[4], 2, 3);
  LAB_WAIT(2, xx[5]); acc.lo += __builtin_shufflevector(xx[5], xx[5], 0, 1); acc.hi += __builtin_shufflevector(xx[5], xx[5], 2, 3);
  LAB_WAIT(1, xx[6]); acc.lo += __builtin_shufflevector(xx[6], xx[6], 0, 1); acc.hi += __builtin_shufflevector(xx[6], xx[6], 2, 3);
  LAB_WAIT(0, xx[7]); acc.lo += __builtin_shufflevector(xx[7], xx[7], 0, 1); acc.hi += __builtin_shufflevector(xx[7], xx[7], 2, 3);
}

__device__ __forceinline__ float4 to_f4(const Acc& a) { return make_float4(a.lo.x, a.lo.y, a.hi.x, a.hi.y); }

// in-kernel finish of a split row (same protocol as spmm_rows_kernel)
__device__ __forceinline__ void finish_split(float4 acc, int row, int slot, int lane, int g, int sub, float4* __restrict__ Y,
                                             float4* __restrict__ partial, const Heavy* __restrict__ heavy,
                                             const int32_t* __restrict__ slot_owner, int32_t* __restrict__ tickets,
                                             const DevEpilogue& ep) {
  constexpr int LPR = 16, G = 4;
  if (g == 0) store_f4_sc1(partial + (size_t)slot * LPR + sub, acc);
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
  float4 sum = f4_zero();
  sum = sum_partials_agent(partial + (size_t)hfirst * LPR + sub, g, G, hn, LPR);
#pragma unroll
  for (int m = LPR; m < 64; m <<= 1) sum = f4_add(sum, f4_shfl_xor(sum, m));
  row_epilogue<LPR>(sum, row, sub, g == 0, Y, ep);
}

// MODE 1: product task / segment records, asm inner loop
//      2: Task64 records (scalar loads) + (col, val) prefetch
//      3: 2, value-free
//      4: 2, column activity from a global bitmap;  5: from an LDS copy of the bitmap
template <int MODE>
__global__ __launch_bounds__(256) void rows_kernel(const Task* __restrict__ tasks, const Task64* __restrict__ tasks64,
                                                   int n_tasks, const Seg* __restrict__ segs,
                                                   const int32_t* __restrict__ indices, const float* __restrict__ vals,
                                                   const float4* __restrict__ X, float4* __restrict__ Y,
                                                   float4* __restrict__ partial, const Heavy* __restrict__ heavy,
                                                   const int32_t* __restrict__ slot_owner, int32_t* __restrict__ tickets,
                                                   const uint32_t* __restrict__ col_bits, int n_bit_words, int pad_row, DevEpilogue ep) {
  constexpr int LPR = 16, G = 4, CH = 64;
  extern __shared__ uint32_t lds_bits[];
  if (MODE == 5) {
    for (int k = threadIdx.x; k < n_bit_words; k += 256) lds_bits[k] = col_bits[k];
    __syncthreads();
  }
  const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256u + threadIdx.x) >> 6));
  if (wave >= n_tasks) return;
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, sub = lane & 15, e16 = sub;
  const unsigned sub16 = (unsigned)sub * 16u;
  const int stamp = ep.mark_stamp ? (int)(*ep.mark_stamp) : 0;
  const floatx4_t zero = {0.f, 0.f, 0.f, 0.f};
  Acc acc = {{0.f, 0.f}, {0.f, 0.f}};
  floatx4_t xx[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) xx[t] = zero;
  unsigned long long t_begin = 0;
  unsigned xcc = 0;
  if (MODE == 60) {
    t_begin = wall_clock64();
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  }
  auto stamp_end = [&](int what) {
    if (MODE == 60 && lane == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      g_probe[3 * (size_t)wave] = t_begin;
      g_probe[3 * (size_t)wave + 1] = wall_clock64();
      g_probe[3 * (size_t)wave + 2] = (xcc & 0xfu) | ((unsigned)what << 8);
    }
  };

  int kind, count, slot0;
  int row, s, e;              // this row-group's row / entry range (coop: the wave's segment)
  if (MODE == 1) {
    const Task tk = tasks[wave];
    kind = __builtin_amdgcn_readfirstlane(tk.kind);
    const int first = __builtin_amdgcn_readfirstlane(tk.first);
    count = __builtin_amdgcn_readfirstlane(tk.count);
    const Seg sg = segs[first + ((kind == 1 && g < count) ? g : 0)];
    row = sg.row; s = sg.start; e = sg.end; slot0 = sg.slot;
  } else {
    const Task64* tp = tasks64 + wave;       // uniform address: scalar loads
    kind = tp->kind; count = tp->count; slot0 = tp->slot;
    const int r0 = tp->row[0], r1 = tp->row[1], r2 = tp->row[2], r3 = tp->row[3];
    const int s0 = tp->start[0], s1 = tp->start[1], s2 = tp->start[2], s3 = tp->start[3];
    const int e0 = tp->end[0], e1 = tp->end[1], e2 = tp->end[2], e3 = tp->end[3];
    const int gg = (kind == 1) ? g : 0;
    row = gg == 0 ? r0 : gg == 1 ? r1 : gg == 2 ? r2 : r3;
    s = gg == 0 ? s0 : gg == 1 ? s1 : gg == 2 ? s2 : s3;
    e = gg == 0 ? e0 : gg == 1 ? e1 : gg == 2 ? e2 : e3;
  }

  // (col, val) of one entry as the gather wants them; col-activity test folded in
  auto fetch = [&](int j, int end, unsigned& cs, float& v) {
    int c = 0;
    v = 0.f;
    if (MODE == 22) { if (j < end) { c = (int)(((unsigned)j * 2654435761u) % (unsigned)pad_row); v = 1.f; } }
    else if (j < end) {
      if (MODE == 23) { c = __builtin_nontemporal_load(indices + j); v = __builtin_nontemporal_load(vals + j); }
      else { c = indices[j]; if (MODE != 3) v = vals[j]; }
    }
    else if (MODE == 3 || MODE == 6) c = pad_row;        // the all-zero row appended to x
    if (MODE == 3) { cs = (unsigned)c << 8; return; }
    if (ep.col_mark && v != 0.f) {
      bool live;
      if (MODE == 4) live = (col_bits[c >> 5] >> (c & 31)) & 1u;
      else if (MODE == 5) live = (lds_bits[c >> 5] >> (c & 31)) & 1u;
      else live = ep.col_mark[c] == stamp;
      if (!live) v = 0.f;
    }
    cs = (unsigned)c << 8;
    if (MODE != 6 && v == 0.f) cs = 0x80000000u;      // padding / dropped edge / dead column: no gather
  };

  if (kind == 0) {
    row = __builtin_amdgcn_readfirstlane(row); s = __builtin_amdgcn_readfirstlane(s); e = __builtin_amdgcn_readfirstlane(e);
    const int slot = __builtin_amdgcn_readfirstlane(slot0);
    if (ep.row_mark && ep.row_mark[row] != stamp) return;
    unsigned cs, csn = 0;
    float v, vn = 0.f;
    fetch(s + 16 * g + e16, e, cs, v);
    for (int base = s; base < e; base += CH) {
      if (MODE >= 2 && base + CH < e) fetch(base + CH + 16 * g + e16, e, csn, vn);     // next chunk in flight
      if (MODE == 3) {
        gather8_novals<false>(cs, sub16, X, xx, acc);
        if (e - base > 8) gather8_novals<true>(cs, sub16, X, xx, acc);
      } else {
        gather8_asm<false, MODE != 6, MODE == 21>(cs, v, sub16, X, xx, acc);
        if (e - base > 8) gather8_asm<true, MODE != 6, MODE == 21>(cs, v, sub16, X, xx, acc);
      }
      if (MODE >= 2) { cs = csn; v = vn; }
      else if (base + CH < e) fetch(base + CH + 16 * g + e16, e, cs, v);
    }
    float4 a4 = to_f4(acc);
#pragma unroll
    for (int m = LPR; m < 64; m <<= 1) a4 = f4_add(a4, f4_shfl_xor(a4, m));
    if (slot < 0) { row_epilogue<LPR>(a4, row, sub, g == 0, Y, ep); stamp_end(0); return; }
    finish_split(a4, row, slot, lane, g, sub, Y, partial, heavy, slot_owner, tickets, ep);
    stamp_end(1);
    return;
  }

  // ---- one short row per row-group ----
  const bool have = g < count;
  const bool live = have && (!ep.row_mark || ep.row_mark[row] == stamp);
  if (!live) e = s;
  int maxlen = e - s;
#pragma unroll
  for (int m = LPR; m < 64; m <<= 1) maxlen = max(maxlen, __shfl_xor(maxlen, m));
  maxlen = __builtin_amdgcn_readfirstlane(maxlen);
  unsigned cs, csn = 0;
  float v, vn = 0.f;
  fetch(s + e16, e, cs, v);
  for (int q = 0; q * 16 < maxlen; ++q) {
    if (MODE >= 2 && (q + 1) * 16 < maxlen) fetch(s + 16 * (q + 1) + e16, e, csn, vn);
    if (MODE == 3) {
      gather8_novals<false>(cs, sub16, X, xx, acc);
      if (maxlen - 16 * q > 8) gather8_novals<true>(cs, sub16, X, xx, acc);
    } else {
      gather8_asm<false, MODE != 6, MODE == 21>(cs, v, sub16, X, xx, acc);
      if (maxlen - 16 * q > 8) gather8_asm<true, MODE != 6, MODE == 21>(cs, v, sub16, X, xx, acc);
    }
    if (MODE >= 2) { cs = csn; v = vn; }
    else if ((q + 1) * 16 < maxlen) fetch(s + 16 * (q + 1) + e16, e, cs, v);
  }
  if (MODE == 20) { if (live) Y[(size_t)row * LPR + sub] = to_f4(acc); }
  else row_epilogue<LPR>(to_f4(acc), row, sub, live, Y, ep);
  stamp_end(2);
}

// Second generation: K tasks per wave (tasks b, b + NB, ... of its workgroup column, so a wave keeps its XCD class),
// every task record read with scalar loads up front, and the FIRST (col, val) chunk of task j+1 in flight under the
// last gathers of task j -- the per-task chain "record -> (col, val) -> gathers -> store" loses its first two links.
// DEPTH 2: all 16 entries of a chunk in flight (xx[16]) instead of 8 + 8.  VALS false: value-free (zero row padding).
// the dense forward flavour only: y + perturbation (counter RNG), one store -- what a flag-specialised instantiation of the
// epilogue would keep live across a multi-task loop
__device__ __forceinline__ void light_epilogue(float4 y, int row, int sub, bool store, float4* __restrict__ Y, const DevEpilogue& ep) {
  const size_t at = (size_t)row * 16 + sub;
  y = perturb_row<16>(y, row, sub, at, nullptr, ep.off_lo, ep.off_hi, ep);
  if (store) Y[at] = y;
}

template <int K, int DEPTH, bool VALS, bool LIGHT = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DEPTH == 2 ? 5 : (LIGHT ? 8 : 7), 8))) void rows_kernel2(const Task64* __restrict__ tasks64, int n_tasks,
                                                    const int32_t* __restrict__ indices, const float* __restrict__ vals,
                                                    const float4* __restrict__ X, float4* __restrict__ Y,
                                                    float4* __restrict__ partial, const Heavy* __restrict__ heavy,
                                                    const int32_t* __restrict__ slot_owner, int32_t* __restrict__ tickets,
                                                    int pad_row, DevEpilogue ep) {
  constexpr int LPR = 16;
  constexpr int NX = (DEPTH == 2 && VALS) ? 16 : 8;
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, sub = lane & 15, e16 = sub;
  const unsigned sub16 = (unsigned)sub * 16u;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nb = (int)gridDim.x;
  const int stamp = ep.mark_stamp ? (int)(*ep.mark_stamp) : 0;
  const floatx4_t zero = {0.f, 0.f, 0.f, 0.f};
  // this lane's view of a task: entry cursor of chunk 0, stride between chunks, end, row; uniform: kind, chunks, slot
  struct View { int kind, n_chunks, slot, rem0; int row, j0, stride, end; bool live; };
  auto view_of = [&](int t) {
    View w{};
    if (t >= n_tasks) { w.n_chunks = 0; w.kind = -1; return w; }
    const Task64* tp = tasks64 + t;
    const int kind = tp->kind, count = tp->count;
    const int r0 = tp->row[0], r1 = tp->row[1], r2 = tp->row[2], r3 = tp->row[3];
    const int s0 = tp->start[0], s1 = tp->start[1], s2 = tp->start[2], s3 = tp->start[3];
    const int e0 = tp->end[0], e1 = tp->end[1], e2 = tp->end[2], e3 = tp->end[3];
    w.kind = kind; w.slot = tp->slot;
    if (kind == 0) {
      w.row = r0; w.end = e0; w.stride = 64; w.j0 = s0 + 16 * g + e16;
      w.live = LIGHT || !ep.row_mark || ep.row_mark[r0] == stamp;
      w.rem0 = w.live ? e0 - s0 : 0;
      w.n_chunks = (w.rem0 + 63) >> 6;
    } else {
      w.row = g == 0 ? r0 : g == 1 ? r1 : g == 2 ? r2 : r3;
      const int s = g == 0 ? s0 : g == 1 ? s1 : g == 2 ? s2 : s3;
      int e = g == 0 ? e0 : g == 1 ? e1 : g == 2 ? e2 : e3;
      w.live = g < count && (LIGHT || !ep.row_mark || ep.row_mark[w.row] == stamp);
      if (!w.live) e = s;
      w.end = e; w.stride = 16; w.j0 = s + e16;
      int maxlen;
      if (LIGHT || !ep.row_mark) {
        maxlen = max(max(e0 - s0, e1 - s1), max(e2 - s2, e3 - s3));          // scalar unit: the record is in SGPRs
      } else {
        maxlen = e - s;
#pragma unroll
        for (int m = LPR; m < 64; m <<= 1) maxlen = max(maxlen, __shfl_xor(maxlen, m));
        maxlen = __builtin_amdgcn_readfirstlane(maxlen);
      }
      w.rem0 = maxlen;
      w.n_chunks = (maxlen + 15) >> 4;
    }
    return w;
  };
  auto fetch = [&](int j, int end, unsigned& cs, float& v) {
    int c = VALS ? 0 : pad_row;
    v = 0.f;
    if (j < end) { c = indices[j]; if (VALS) v = vals[j]; }
    if (!LIGHT && VALS && ep.col_mark && v != 0.f && ep.col_mark[c] != stamp) v = 0.f;
    cs = (unsigned)c << 8;
    if (VALS && v == 0.f) cs = 0x80000000u;
  };

  // first chunk of a task as seen WITHOUT its row marks (the full view is derived when the task starts: carrying a
  // second View through the chunk loop cost 36 VGPRs and 57 spilled SGPRs)
  auto peek = [&](int t, int& j0, int& end) {
    j0 = 0; end = 0;
    if (t >= n_tasks) return false;
    const Task64* tp = tasks64 + t;
    const int kind = tp->kind, count = tp->count;
    if (kind == 0) { j0 = tp->start[0] + 16 * g + e16; end = tp->end[0]; return true; }
    const int s = g == 0 ? tp->start[0] : g == 1 ? tp->start[1] : g == 2 ? tp->start[2] : tp->start[3];
    const int e = g == 0 ? tp->end[0] : g == 1 ? tp->end[1] : g == 2 ? tp->end[2] : tp->end[3];
    j0 = s + e16; end = g < count ? e : s;
    return true;
  };

  unsigned cs = 0x80000000u, csn = 0x80000000u;
  float v = 0.f, vn = 0.f;
  {
    int j0, end;
    if (peek((int)blockIdx.x * 4 + wv, j0, end)) fetch(j0, end, cs, v);
  }
#pragma unroll 1
  for (int k = 0; k < K; ++k) {
    const int t = ((int)blockIdx.x + k * nb) * 4 + wv;
    if (t >= n_tasks) break;
    const View cur = view_of(t);
    if (!cur.live && cur.kind == 1 && VALS) cs = 0x80000000u;      // prefetched entries of a dead row: no gathers
    int nj0 = 0, nend = 0;
    const bool has_next = (k + 1 < K) && peek(t + 4 * nb, nj0, nend);
    Acc acc = {{0.f, 0.f}, {0.f, 0.f}};
    floatx4_t xx[NX];            // (per task, so that the gather registers are dead during the epilogue)
#pragma unroll
    for (int t = 0; t < NX; ++t) xx[t] = zero;
    for (int q = 0; q < cur.n_chunks; ++q) {
      if (q + 1 < cur.n_chunks) fetch(cur.j0 + cur.stride * (q + 1), cur.end, csn, vn);     // next chunk of this task
      else if (has_next) fetch(nj0, nend, csn, vn);                                          // first chunk of the next
      const int rem = cur.rem0 - (cur.kind == 0 ? 64 : 16) * q;
      if constexpr (!VALS) {
        gather8_novals<false>(cs, sub16, X, xx, acc);
        if (rem > 8) gather8_novals<true>(cs, sub16, X, xx, acc);
      } else if constexpr (DEPTH == 2) {
        if (rem > 8) gather16_asm(cs, v, sub16, X, xx, acc);
        else gather8_lo16(cs, v, sub16, X, xx, acc);
      } else {
        gather8_asm<false, true>(cs, v, sub16, X, xx, acc);
        if (rem > 8) gather8_asm<true, true>(cs, v, sub16, X, xx, acc);
      }
      cs = csn; v = vn;
    }
    if (cur.n_chunks == 0 && has_next) fetch(nj0, nend, cs, v);       // (whole task dead: nothing was prefetched)
    // ---- epilogue of task k
    if (cur.kind == 0) {
      if (cur.live) {
        const int row = __builtin_amdgcn_readfirstlane(cur.row), slot = __builtin_amdgcn_readfirstlane(cur.slot);
        float4 a4 = to_f4(acc);
#pragma unroll
        for (int m = LPR; m < 64; m <<= 1) a4 = f4_add(a4, f4_shfl_xor(a4, m));
        if (slot < 0) {
          if (LIGHT) light_epilogue(a4, row, sub, g == 0, Y, ep); else row_epilogue<LPR>(a4, row, sub, g == 0, Y, ep);
        } else {
          finish_split(a4, row, slot, lane, g, sub, Y, partial, heavy, slot_owner, tickets, ep);
        }
      }
    } else {
      if (LIGHT) light_epilogue(to_f4(acc), cur.row, sub, cur.live, Y, ep);
      else row_epilogue<LPR>(to_f4(acc), cur.row, sub, cur.live, Y, ep);
    }
  }
}

// Third generation: a ROLLING window of eight gathers.  Variant 2 issues 8 loads, drains them, issues the next 8: two
// memory round trips per 16-entry chunk, and between them the wave has nothing in flight.  Here entry t of batch b+1 is
// issued into the register entry t of batch b has just been consumed from, so eight rows stay in flight from the first
// batch of a row to its last (across chunk boundaries too: the next chunk's (col, val) are prefetched).
// (two statements: with the halves of xx[T] as inputs AND xx[T] as the tied load destination in ONE asm, the compiler
// copies the halves to fresh registers -- 156 VGPRs)
#define LAB_ROLL(T, TN, VP, SEL, SRC)                                                                              \
  LAB_FMA(7, SEL, VP, xx[T]);                                                                                      \
  asm volatile("s_nop 1\n\t"                                                                                       \
               "v_or_b32_dpp %[of], %[cs], %[s16] row_newbcast:" #TN " row_mask:0xf bank_mask:0xf\n\t"              \
               "s_mov_b64 %[sv], exec\n\t"                                                                         \
               "v_cmpx_le_i32_e32 0, %[of]\n\t"                                                                    \
               "global_load_dwordx4 %[x], %[of], %[b]\n\t"                                                         \
               "s_mov_b64 exec, %[sv]"                                                                             \
               : [x] "+v"(xx[T]), [of] "=&v"(otmp), [sv] "=&s"(stmp)                                               \
               : [cs] "v"(SRC), [s16] "v"(sub16), [b] "s"(X)                                                       \
               : "memory", "vcc")

// consume batch (values VLO/VHI half of `v`) while issuing the next batch from `src` (half NEXT_HI)
template <bool CUR_HI, bool NEXT_HI>
__device__ __forceinline__ void roll8(float v, unsigned src, unsigned sub16, const void* X, floatx4_t (&xx)[8], Acc& acc) {
  float vv[8];
  unsigned otmp;
  unsigned long long stmp;
  asm volatile("s_nop 4" : "+v"(v), "+v"(src));     // (exec was last written by a v_cmpx: 5 wait states before a DPP op)
  if (!CUR_HI) {
    LAB_DPP_MOV(0); LAB_DPP_MOV(1); LAB_DPP_MOV(2); LAB_DPP_MOV(3); LAB_DPP_MOV(4); LAB_DPP_MOV(5); LAB_DPP_MOV(6); LAB_DPP_MOV(7);
  } else {
    LAB_DPP_MOV(8); LAB_DPP_MOV(9); LAB_DPP_MOV(10); LAB_DPP_MOV(11); LAB_DPP_MOV(12); LAB_DPP_MOV(13); LAB_DPP_MOV(14); LAB_DPP_MOV(15);
  }
  const floatx2_t p0 = {vv[0], vv[1]}, p1 = {vv[2], vv[3]}, p2 = {vv[4], vv[5]}, p3 = {vv[6], vv[7]};
  if (!NEXT_HI) {
    LAB_ROLL(0, 0, p0, 0, src); LAB_ROLL(1, 1, p0, 1, src); LAB_ROLL(2, 2, p1, 0, src); LAB_ROLL(3, 3, p1, 1, src);
    LAB_ROLL(4, 4, p2, 0, src); LAB_ROLL(5, 5, p2, 1, src); LAB_ROLL(6, 6, p3, 0, src); LAB_ROLL(7, 7, p3, 1, src);
  } else {
    LAB_ROLL(0, 8, p0, 0, src); LAB_ROLL(1, 9, p0, 1, src); LAB_ROLL(2, 10, p1, 0, src); LAB_ROLL(3, 11, p1, 1, src);
    LAB_ROLL(4, 12, p2, 0, src); LAB_ROLL(5, 13, p2, 1, src); LAB_ROLL(6, 14, p3, 0, src); LAB_ROLL(7, 15, p3, 1, src);
  }
}
// the last batch of a row: consume only
template <bool CUR_HI>
__device__ __forceinline__ void drain8(float v, const void* X, floatx4_t (&xx)[8], Acc& acc) {
  float vv[8];
  asm volatile("s_nop 4" : "+v"(v));
  if (!CUR_HI) {
    LAB_DPP_MOV(0); LAB_DPP_MOV(1); LAB_DPP_MOV(2); LAB_DPP_MOV(3); LAB_DPP_MOV(4); LAB_DPP_MOV(5); LAB_DPP_MOV(6); LAB_DPP_MOV(7);
  } else {
    LAB_DPP_MOV(8); LAB_DPP_MOV(9); LAB_DPP_MOV(10); LAB_DPP_MOV(11); LAB_DPP_MOV(12); LAB_DPP_MOV(13); LAB_DPP_MOV(14); LAB_DPP_MOV(15);
  }
  const floatx2_t p0 = {vv[0], vv[1]}, p1 = {vv[2], vv[3]}, p2 = {vv[4], vv[5]}, p3 = {vv[6], vv[7]};
  LAB_FMA(7, 0, p0, xx[0]); LAB_FMA(6, 1, p0, xx[1]); LAB_FMA(5, 0, p1, xx[2]); LAB_FMA(4, 1, p1, xx[3]);
  LAB_FMA(3, 0, p2, xx[4]); LAB_FMA(2, 1, p2, xx[5]); LAB_FMA(1, 0, p3, xx[6]); LAB_FMA(0, 1, p3, xx[7]);
}
// the first batch of a row: issue only (entries 0-7 of `cs`)
__device__ __forceinline__ void issue8(unsigned cs, unsigned sub16, const void* X, floatx4_t (&xx)[8]) {
  unsigned off[8];
  asm volatile("s_nop 4" : "+v"(cs));
  LAB_DPP_OR(0); LAB_DPP_OR(1); LAB_DPP_OR(2); LAB_DPP_OR(3); LAB_DPP_OR(4); LAB_DPP_OR(5); LAB_DPP_OR(6); LAB_DPP_OR(7);
  pred_load8(xx[0], xx[1], xx[2], xx[3], xx[4], xx[5], xx[6], xx[7], off, X);
}

__global__ __launch_bounds__(256) void rows_kernel3(const Task64* __restrict__ tasks64, int n_tasks,
                                                    const int32_t* __restrict__ indices, const float* __restrict__ vals,
                                                    const float4* __restrict__ X, float4* __restrict__ Y,
                                                    float4* __restrict__ partial, const Heavy* __restrict__ heavy,
                                                    const int32_t* __restrict__ slot_owner, int32_t* __restrict__ tickets,
                                                    DevEpilogue ep) {
  constexpr int LPR = 16;
  const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256u + threadIdx.x) >> 6));
  if (wave >= n_tasks) return;
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, sub = lane & 15, e16 = sub;
  const unsigned sub16 = (unsigned)sub * 16u;
  const int stamp = ep.mark_stamp ? (int)(*ep.mark_stamp) : 0;
  const floatx4_t zero = {0.f, 0.f, 0.f, 0.f};
  Acc acc = {{0.f, 0.f}, {0.f, 0.f}};
  floatx4_t xx[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) xx[t] = zero;
  const Task64* tp = tasks64 + wave;
  const int kind = tp->kind, count = tp->count, slot = tp->slot;
  int row = tp->row[0], s = tp->start[0], e = tp->end[0];
  if (kind == 1) {
    const int r1 = tp->row[1], s1 = tp->start[1], e1 = tp->end[1], r2 = tp->row[2], s2 = tp->start[2], e2 = tp->end[2];
    const int r3 = tp->row[3], s3 = tp->start[3], e3 = tp->end[3];
    if (g == 1) { row = r1; s = s1; e = e1; }
    if (g == 2) { row = r2; s = s2; e = e2; }
    if (g == 3) { row = r3; s = s3; e = e3; }
  }
  auto fetch = [&](int j, int end, unsigned& cs, float& v) {
    int c = 0;
    v = 0.f;
    if (j < end) { c = indices[j]; v = vals[j]; }
    cs = (v == 0.f) ? 0x80000000u : (unsigned)c << 8;
  };
  // per-lane cursor of chunk 0, stride between chunks; uniform entry count of the longest row-group
  int j0, stride, total;
  bool live = true;
  if (kind == 0) {
    row = __builtin_amdgcn_readfirstlane(row); s = __builtin_amdgcn_readfirstlane(s); e = __builtin_amdgcn_readfirstlane(e);
    if (ep.row_mark && ep.row_mark[row] != stamp) return;
    j0 = s + 16 * g + e16; stride = 64;
    // a coop chunk gives every row-group 16 entries; the last chunk may give the later groups fewer (or none)
    total = ((e - s) >> 6) * 16 + min(16, (e - s) & 63);      // entries of row-group 0 (the longest)
  } else {
    live = g < count && (!ep.row_mark || ep.row_mark[row] == stamp);
    if (!live) e = s;
    int maxlen = e - s;
#pragma unroll
    for (int m = LPR; m < 64; m <<= 1) maxlen = max(maxlen, __shfl_xor(maxlen, m));
    total = __builtin_amdgcn_readfirstlane(maxlen);
    j0 = s + e16; stride = 16;
  }
  // straight-line body (no special first / last batch: an exhausted half carries sign-bit offsets and v = 0, its
  // loads are exec = 0 no-ops that still count in vmcnt, so every wait below stays exact)
  const int nchunks = (total + 15) >> 4;
  if (nchunks > 0) {
    unsigned cs, csn = 0x80000000u;
    float v, vn = 0.f;
    fetch(j0, e, cs, v);
    if (nchunks > 1) fetch(j0 + stride, e, csn, vn);
    issue8(cs, sub16, X, xx);
    for (int q = 0; q < nchunks; ++q) {
      roll8<false, true>(v, cs, sub16, X, xx, acc);       // consume entries 0-7 of chunk q, issue its entries 8-15
      roll8<true, false>(v, csn, sub16, X, xx, acc);      // consume entries 8-15, issue entries 0-7 of chunk q + 1
      cs = csn; v = vn;
      csn = 0x80000000u; vn = 0.f;
      if (q + 2 < nchunks) fetch(j0 + stride * (q + 2), e, csn, vn);
    }
  }
  if (kind == 0) {
    float4 a4 = to_f4(acc);
#pragma unroll
    for (int m = LPR; m < 64; m <<= 1) a4 = f4_add(a4, f4_shfl_xor(a4, m));
    if (slot < 0) { row_epilogue<LPR>(a4, row, sub, g == 0, Y, ep); return; }
    finish_split(a4, row, slot, lane, g, sub, Y, partial, heavy, slot_owner, tickets, ep);
    return;
  }
  row_epilogue<LPR>(to_f4(acc), row, sub, live, Y, ep);
}

// Fifth generation: PERSISTENT waves (2048 workgroups = 8 waves per SIMD, one round).  Wave w first runs the cooperative
// tasks w, w + W, ... exactly as the product does, then walks its short-row tasks base + w, base + w + W, ... with the NEXT
// task's record (scalar) and first (col, val) chunk in flight under the current task's last gathers -- the two loops are
// separate so that neither carries the other's state.  `n_coop_pad`: cooperative tasks padded to a multiple of 32 with
// empty records, so every task of a wave has the wave's XCD class.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 8)))
void rows_kernel5(const Task64* __restrict__ tasks64, int n_coop_pad, int n_tasks,
                  const int32_t* __restrict__ indices, const float* __restrict__ vals,
                  const float4* __restrict__ X, float4* __restrict__ Y, float4* __restrict__ partial,
                  const Heavy* __restrict__ heavy, const int32_t* __restrict__ slot_owner, int32_t* __restrict__ tickets,
                  DevEpilogue ep) {
  constexpr int LPR = 16;
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, sub = lane & 15, e16 = sub;
  const unsigned sub16 = (unsigned)sub * 16u;
  const int w0 = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256u + threadIdx.x) >> 6));
  const int W = (int)gridDim.x * 4;
  const int stamp = ep.mark_stamp ? (int)(*ep.mark_stamp) : 0;
  const floatx4_t zero = {0.f, 0.f, 0.f, 0.f};
  auto fetch = [&](int j, int end, unsigned& cs, float& v) {
    int c = 0;
    v = 0.f;
    if (j < end) { c = indices[j]; v = vals[j]; }
    cs = (v == 0.f) ? 0x80000000u : (unsigned)c << 8;
  };
  // ---- cooperative tasks (long rows / split segments): one per iteration, no cross-task pipeline
#pragma unroll 1
  for (int t = w0; t < n_coop_pad; t += W) {
    const Task64* tp = tasks64 + t;
    if (tp->kind != 0) continue;                          // padding record
    const int row = tp->row[0], s = tp->start[0], e = tp->end[0], slot = tp->slot;
    if (ep.row_mark && ep.row_mark[row] != stamp) continue;
    Acc acc = {{0.f, 0.f}, {0.f, 0.f}};
    floatx4_t xx[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) xx[k] = zero;
    unsigned cs, csn = 0x80000000u;
    float v, vn = 0.f;
    fetch(s + 16 * g + e16, e, cs, v);
    for (int base = s; base < e; base += 64) {
      if (base + 64 < e) fetch(base + 64 + 16 * g + e16, e, csn, vn);
      gather8_asm<false, true>(cs, v, sub16, X, xx, acc);
      if (e - base > 8) gather8_asm<true, true>(cs, v, sub16, X, xx, acc);
      cs = csn; v = vn;
    }
    float4 a4 = to_f4(acc);
#pragma unroll
    for (int m = LPR; m < 64; m <<= 1) a4 = f4_add(a4, f4_shfl_xor(a4, m));
    if (slot < 0) row_epilogue<LPR>(a4, row, sub, g == 0, Y, ep);
    else finish_split(a4, row, slot, lane, g, sub, Y, partial, heavy, slot_owner, tickets, ep);
  }
  // ---- short rows: four per task, next task prefetched
  int t = n_coop_pad + w0;
  if (t >= n_tasks) return;
  auto lane_view = [&](const Task64* tp, int& row, int& s, int& e) {
    const int count = tp->count;
    row = g == 0 ? tp->row[0] : g == 1 ? tp->row[1] : g == 2 ? tp->row[2] : tp->row[3];
    s = g == 0 ? tp->start[0] : g == 1 ? tp->start[1] : g == 2 ? tp->start[2] : tp->start[3];
    e = g == 0 ? tp->end[0] : g == 1 ? tp->end[1] : g == 2 ? tp->end[2] : tp->end[3];
    if (g >= count) e = s;
  };
  unsigned cs, csn = 0x80000000u;
  float v, vn = 0.f;
  int row, s, e;
  lane_view(tasks64 + t, row, s, e);
  fetch(s + e16, e, cs, v);
#pragma unroll 1
  for (;;) {
    const bool live = e > s && (!ep.row_mark || ep.row_mark[row] == stamp);
    if (!live) { e = s; cs = 0x80000000u; }
    int maxlen = e - s;
#pragma unroll
    for (int m = LPR; m < 64; m <<= 1) maxlen = max(maxlen, __shfl_xor(maxlen, m));
    maxlen = __builtin_amdgcn_readfirstlane(maxlen);
    const int tn = t + W;
    int nrow = 0, ns = 0, ne = 0;
    if (tn < n_tasks) lane_view(tasks64 + tn, nrow, ns, ne);
    Acc acc = {{0.f, 0.f}, {0.f, 0.f}};
    floatx4_t xx[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) xx[k] = zero;
    const int nq = (maxlen + 15) >> 4;
    for (int q = 0; q < nq; ++q) {
      if (q + 1 < nq) fetch(s + 16 * (q + 1) + e16, e, csn, vn);
      else fetch(ns + e16, ne, csn, vn);                         // first chunk of the next task (empty when none)
      gather8_asm<false, true>(cs, v, sub16, X, xx, acc);
      if (maxlen - 16 * q > 8) gather8_asm<true, true>(cs, v, sub16, X, xx, acc);
      cs = csn; v = vn;
    }
    if (nq == 0) fetch(ns + e16, ne, cs, v);
    row_epilogue<LPR>(to_f4(acc), row, sub, live, Y, ep);
    if (tn >= n_tasks) break;
    t = tn; row = nrow; s = ns; e = ne;
  }
}

// marks -> bitmap (1 = column live this step)
__global__ void build_bits(const int32_t* __restrict__ mark, const int64_t* __restrict__ stamp, int n, uint32_t* __restrict__ bits) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w * 32 >= n) return;
  const int st = (int)*stamp;
  uint32_t b = 0;
  for (int k = 0; k < 32 && w * 32 + k < n; ++k) b |= (mark[w * 32 + k] == st ? 1u : 0u) << k;
  bits[w] = b;
}

struct Lab {
  Task64* d_tasks64 = nullptr;
  int n_tasks = 0;
  Task64* d_tasks64p = nullptr;     // cooperative tasks padded to a multiple of 32, then the short-row tasks
  int n_coop_pad = 0, n_tasks_p = 0;
  uint32_t* d_bits = nullptr;
  int n_bit_words = 0;
};

}  // namespace lab

extern "C" {

// Task64 records for the plan's 4-rows-per-wave task list (LPR = 16)
int lab_create(void** out, const srh_spmm_plan_t* plan) {
  using namespace lab;
  const int n = plan->n_tasks[1];
  std::vector<Task> tasks(n);
  if (hipMemcpy(tasks.data(), plan->d_tasks[1], sizeof(Task) * n, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  int n_segs = 0;
  for (const Task& t : tasks) n_segs = std::max(n_segs, t.first + std::max(1, t.count));
  std::vector<Seg> segs(n_segs);
  if (hipMemcpy(segs.data(), plan->d_tsegs, sizeof(Seg) * n_segs, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  std::vector<Task64> t64(n);
  for (int k = 0; k < n; ++k) {
    Task64 r{};
    r.kind = tasks[k].kind; r.count = tasks[k].count; r.slot = -1;
    for (int q = 0; q < 4; ++q) {
      const Seg& sg = segs[tasks[k].first + ((tasks[k].kind == 1 && q < tasks[k].count) ? q : 0)];
      r.row[q] = sg.row; r.start[q] = sg.start; r.end[q] = (tasks[k].kind == 1 && q >= tasks[k].count) ? sg.start : sg.end;
      if (q == 0) r.slot = sg.slot;
    }
    t64[k] = r;
  }
  Lab* L = new Lab();
  L->n_tasks = n;
  L->n_bit_words = (int)((plan->n_cols + 31) / 32);
  if (hipMalloc(&L->d_tasks64, sizeof(Task64) * std::max(1, n)) != hipSuccess) return -1;
  if (hipMemcpy(L->d_tasks64, t64.data(), sizeof(Task64) * n, hipMemcpyHostToDevice) != hipSuccess) return -1;
  {
    std::vector<Task64> padded;
    size_t k = 0;
    for (; k < t64.size() && t64[k].kind == 0; ++k) padded.push_back(t64[k]);
    Task64 empty{};
    empty.kind = 1; empty.count = 0;
    while (padded.size() % 32) padded.push_back(empty);
    L->n_coop_pad = (int)padded.size();
    for (; k < t64.size(); ++k) { if (t64[k].kind != 1) return -5; padded.push_back(t64[k]); }
    L->n_tasks_p = (int)padded.size();
    if (hipMalloc(&L->d_tasks64p, sizeof(Task64) * padded.size()) != hipSuccess) return -1;
    if (hipMemcpy(L->d_tasks64p, padded.data(), sizeof(Task64) * padded.size(), hipMemcpyHostToDevice) != hipSuccess) return -1;
  }
  if (hipMalloc(&L->d_bits, sizeof(uint32_t) * L->n_bit_words) != hipSuccess) return -1;
  if (hipMemset(L->d_bits, 0, sizeof(uint32_t) * L->n_bit_words) != hipSuccess) return -1;
  *out = L;
  return 0;
}

void lab_destroy(void* h) {
  lab::Lab* L = reinterpret_cast<lab::Lab*>(h);
  if (!L) return;
  (void)hipFree(L->d_tasks64); (void)hipFree(L->d_bits);
  delete L;
}

// where variant 60 writes its 3 x n_tasks records (uint64: begin, end, xcd | exit << 8)
int lab_set_probe(void* d_buf) {
  unsigned long long* p = reinterpret_cast<unsigned long long*>(d_buf);
  return hipMemcpyToSymbol(HIP_SYMBOL(lab::g_probe), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
int lab_n_tasks(void* h) { return reinterpret_cast<lab::Lab*>(h)->n_tasks; }

// ---- custom task lists (run.py --balance): the plan's Task64 records on the host, and launches over a caller's list ----
// The plan binds task k to block k / 4 and the hardware binds block b to XCD b % 8, so every XCD gets the same NUMBER of
// blocks; the XCDs do not take the same TIME over them (profiles/r02_i_wave_timeline_*: 3.5 us between the first and the
// last to finish).  A list in which a slow XCD's last blocks are EMPTY records (kind 1, count 0: the wave leaves at once)
// and their tasks sit in extra blocks of a fast XCD gives unequal shares without touching the kernel.
int lab_get_tasks64(const srh_spmm_plan_t* plan, void* host_out, int max_n) {
  const int n = plan->n_tasks[1];
  if (n > max_n) return -n;
  return hipMemcpy(host_out, plan->d_tasks64[1], sizeof(Task64) * (size_t)n, hipMemcpyDeviceToHost) == hipSuccess ? n : -1;
}
// probe = 0: the PRODUCT kernel over d_tasks (n_tasks records, a multiple of 4);  probe = 1: the stamping kernel
// (rows_kernel<60>: begin / end / XCD of every wave into the buffer set by lab_set_probe, 3 x n_tasks uint64)
int lab_spmm_custom(void* h, const srh_spmm_plan_t* plan, const void* d_tasks, int n_tasks, const int32_t* d_indices,
                    const float* d_vals, const float* d_x, float* d_y, const srh_spmm_epilogue_t* epi, void* stream, int probe) {
  using namespace lab;
  Lab* L = reinterpret_cast<Lab*>(h);
  DevEpilogue ep{};
  if (translate_epilogue(epi, 64, d_x, d_y, ep) != SRH_OK) return -2;
  if (ep.col_mark) return -6;
  hipStream_t st = srh::as_stream(stream);
  const Task64* tasks = reinterpret_cast<const Task64*>(d_tasks);
  const int blocks = (n_tasks + 3) / 4;
  if (probe) {
    if (!d_vals) return -7;
    rows_kernel<60><<<blocks, 256, 0, st>>>(plan->d_tasks[1], tasks, n_tasks, plan->d_tsegs, d_indices, d_vals,
                                            reinterpret_cast<const float4*>(d_x), reinterpret_cast<float4*>(d_y),
                                            reinterpret_cast<float4*>(plan->d_partial), plan->d_heavy, plan->d_slot_owner,
                                            plan->d_tickets, L->d_bits, L->n_bit_words, (int)plan->n_cols, ep);
  } else {
    srh_batch_fetch_args_t no_rider{};
    spmm_rows_kernel<16, false><<<blocks, 256, 0, st>>>(
        tasks, n_tasks, d_indices, d_vals, reinterpret_cast<const float4*>(d_x), reinterpret_cast<float4*>(d_y),
        reinterpret_cast<float4*>(plan->d_partial), plan->d_heavy, plan->d_slot_owner, plan->d_tickets, ep, 0, no_rider);
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// refresh the column bitmap from the epilogue's col_mark / stamp (not timed with the product: batch_fetch would write it)
int lab_build_bits(void* h, const int32_t* d_mark, const int64_t* d_stamp, int n, void* stream) {
  lab::Lab* L = reinterpret_cast<lab::Lab*>(h);
  const int words = (n + 31) / 32;
  lab::build_bits<<<(words + 255) / 256, 256, 0, srh::as_stream(stream)>>>(d_mark, d_stamp, n, L->d_bits);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int lab_spmm(void* h, const srh_spmm_plan_t* plan, const int32_t* d_indices, const float* d_vals, const float* d_x,
             float* d_y, const srh_spmm_epilogue_t* epi, void* stream, int variant) {
  using namespace lab;
  Lab* L = reinterpret_cast<Lab*>(h);
  DevEpilogue ep{};
  if (translate_epilogue(epi, 64, d_x, d_y, ep) != SRH_OK) return -2;
  hipStream_t st = srh::as_stream(stream);
  const int n = plan->n_tasks[1];
  const int blocks = (n + 3) / 4;
#define LAB_LAUNCH(M, SH)                                                                                           \
  rows_kernel<M><<<blocks, 256, SH, st>>>(plan->d_tasks[1], L->d_tasks64, n, plan->d_tsegs, d_indices, d_vals,       \
                                          reinterpret_cast<const float4*>(d_x), reinterpret_cast<float4*>(d_y),      \
                                          reinterpret_cast<float4*>(plan->d_partial), plan->d_heavy,                 \
                                          plan->d_slot_owner, plan->d_tickets, L->d_bits, L->n_bit_words, (int)plan->n_cols, ep)
  switch (variant) {
    case 1: LAB_LAUNCH(1, 0); break;
    case 2: LAB_LAUNCH(2, 0); break;
    case 3: LAB_LAUNCH(3, 0); break;
    case 4: LAB_LAUNCH(4, 0); break;
    case 5: LAB_LAUNCH(5, L->n_bit_words * 4); break;
    case 6: LAB_LAUNCH(6, 0); break;
    case 20: LAB_LAUNCH(20, 0); break;
    case 23: LAB_LAUNCH(23, 0); break;
    case 21: LAB_LAUNCH(21, 0); break;
    case 22: LAB_LAUNCH(22, 0); break;
    case 60: LAB_LAUNCH(60, 0); break;
#define LAB_PRODUCT(VALS)                                                                                             \
  do {                                                                                                                \
    if (ep.col_mark) return -6;                                                                                       \
    srh_batch_fetch_args_t no_rider{};                                                                                \
    spmm_rows_kernel<16, false><<<blocks, 256, 0, st>>>(                                                              \
        plan->d_tasks64[1], n, d_indices, VALS, reinterpret_cast<const float4*>(d_x), reinterpret_cast<float4*>(d_y), \
        reinterpret_cast<float4*>(plan->d_partial), plan->d_heavy, plan->d_slot_owner, plan->d_tickets, ep, 0, no_rider); \
  } while (0)
    case 75: LAB_PRODUCT(d_vals); break;          /* the product kernel launched from here (canonical task list) */
    case 76: LAB_PRODUCT(nullptr); break;         /* ... as a pattern product (no value stream): vs all-ones values */
#define LAB_LAUNCH2(KK, DD, VV)                                                                                      \
  do {                                                                                                               \
    int nblk = (blocks + KK - 1) / KK;                                                                               \
    nblk = (nblk + 7) / 8 * 8; /* a wave's tasks b, b + NB, ... keep one XCD class when NB % 8 == 0 */               \
    rows_kernel2<KK, DD, VV><<<nblk, 256, 0, st>>>(L->d_tasks64, n, d_indices, d_vals,                                \
                                                   reinterpret_cast<const float4*>(d_x), reinterpret_cast<float4*>(d_y), \
                                                   reinterpret_cast<float4*>(plan->d_partial), plan->d_heavy,        \
                                                   plan->d_slot_owner, plan->d_tickets, (int)plan->n_cols, ep);      \
  } while (0)
    case 30:
      rows_kernel3<<<blocks, 256, 0, st>>>(L->d_tasks64, n, d_indices, d_vals, reinterpret_cast<const float4*>(d_x),
                                           reinterpret_cast<float4*>(d_y), reinterpret_cast<float4*>(plan->d_partial),
                                           plan->d_heavy, plan->d_slot_owner, plan->d_tickets, ep);
      break;
    case 50:
      rows_kernel5<<<2048, 256, 0, st>>>(L->d_tasks64p, L->n_coop_pad, L->n_tasks_p, d_indices, d_vals,
                                         reinterpret_cast<const float4*>(d_x), reinterpret_cast<float4*>(d_y),
                                         reinterpret_cast<float4*>(plan->d_partial), plan->d_heavy, plan->d_slot_owner,
                                         plan->d_tickets, ep);
      break;
    case 40: {
      const int nblk = 2048;                      /* 8 waves per SIMD x 1024 SIMDs, one round: persistent */
      const int kk = (blocks + nblk - 1) / nblk;
      if (kk <= 3) rows_kernel2<3, 1, true, true><<<nblk, 256, 0, st>>>(L->d_tasks64, n, d_indices, d_vals,
          reinterpret_cast<const float4*>(d_x), reinterpret_cast<float4*>(d_y), reinterpret_cast<float4*>(plan->d_partial),
          plan->d_heavy, plan->d_slot_owner, plan->d_tickets, (int)plan->n_cols, ep);
      else return -4;
      break;
    }
    case 41:
      rows_kernel2<1, 1, true, true><<<blocks, 256, 0, st>>>(L->d_tasks64, n, d_indices, d_vals,
          reinterpret_cast<const float4*>(d_x), reinterpret_cast<float4*>(d_y), reinterpret_cast<float4*>(plan->d_partial),
          plan->d_heavy, plan->d_slot_owner, plan->d_tickets, (int)plan->n_cols, ep);
      break;
    case 10: LAB_LAUNCH2(1, 1, true); break;
    case 11: LAB_LAUNCH2(2, 1, true); break;
    case 12: LAB_LAUNCH2(3, 1, true); break;
    case 13: LAB_LAUNCH2(4, 1, true); break;
    case 14: LAB_LAUNCH2(1, 2, true); break;
    case 15: LAB_LAUNCH2(2, 2, true); break;
    case 16: LAB_LAUNCH2(2, 1, false); break;
    case 17: LAB_LAUNCH2(3, 1, false); break;
    case 18: LAB_LAUNCH2(6, 1, true); break;
    default: return -3;
  }
#undef LAB_LAUNCH
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // extern "C"
