// Does VALU work overlap v_mfma_f32_16x16x4_f32 on gfx950 -- inside one wave, and between the two waves of a SIMD?
// (round 5: nce_tile_f32 showed the matrix pipe 47 % busy with 27 k MFMA + 15 k VALU cycles per SIMD in a 57 k-cycle
//  kernel, i.e. about their SUM).  hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_f32_valu.hip -o /tmp/mv && /tmp/mv
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));

// MODE 0: MFMA only (4 independent accumulators)   1: VALU only (v_fma chains)   2: same wave, 1 MFMA : V VALU interleaved
// MODE 3: waves 0-3 MFMA only, waves 4-7 VALU only (partners on one SIMD)        4: MFMA only, 1 wave per SIMD
template <int MODE, int V>
__global__ __launch_bounds__(512) void k(float* out, int iters, float a, float b) {
  const int wv = threadIdx.x >> 6;
  floatx4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (floatx4){0.f, 0.f, 0.f, 0.f};
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = a + i + threadIdx.x;
  const bool do_mfma = MODE == 0 || MODE == 2 || MODE == 4 || (MODE == 3 && wv < 4);
  const bool do_valu = MODE == 1 || MODE == 2 || (MODE == 3 && wv >= 4);
  if (MODE == 4 && wv >= 4) return;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      if (do_mfma) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 3], 0, 0, 0);
      if (do_valu) {
#pragma unroll
        for (int v = 0; v < V; ++v) x[v & 7] = __builtin_fmaf(x[v & 7], a, b);
      }
      if (MODE == 2) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, V, 0);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += x[i];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int MODE, int V>
void run(const char* what, float* d, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE, V><<<256, 512>>>(d, 10, 1.0f, 0.5f);
  hipEventRecord(e0);
  k<MODE, V><<<256, 512>>>(d, iters, 1.0f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double per16 = ms * 1e6 / iters;   // ns per 16-MFMA group (per wave)
  printf("%-62s V=%2d  %8.1f us   %7.1f ns per 16-MFMA group  (%5.1f ns per MFMA slot)\n", what, V, ms * 1e3, per16, per16 / 16);
}

int main() {
  float* d;
  hipMalloc(&d, 4096);
  const int iters = 2000;
  run<0, 0>("MFMA only, 2 waves/SIMD (both MFMA)", d, iters);
  run<4, 0>("MFMA only, 1 wave/SIMD", d, iters);
  run<1, 8>("VALU only (8 v_fma per slot), 2 waves/SIMD", d, iters);
  run<1, 4>("VALU only (4 v_fma per slot), 2 waves/SIMD", d, iters);
  run<2, 2>("same wave: 1 MFMA + 2 VALU per slot, 2 waves/SIMD", d, iters);
  run<2, 4>("same wave: 1 MFMA + 4 VALU per slot, 2 waves/SIMD", d, iters);
  run<2, 8>("same wave: 1 MFMA + 8 VALU per slot, 2 waves/SIMD", d, iters);
  run<3, 4>("partner waves: one MFMA only, one VALU only (4 per slot)", d, iters);
  run<3, 8>("partner waves: one MFMA only, one VALU only (8 per slot)", d, iters);
  run<3, 16>("partner waves: one MFMA only, one VALU only (16 per slot)", d, iters);
  return 0;
}
