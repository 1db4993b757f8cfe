// Gather roofline of MI355X for 256-byte rows: what the SpMM's inner loop can reach at best.
// Each 16-lane group reads NL random rows (one float4 per lane) per iteration, NL in flight.
// build: hipcc --offload-arch=gfx950 -O3 tools/microbench/gather_bw.hip -o gpurun_out/gather_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int NL>
__global__ __launch_bounds__(256) void gather(const float4* __restrict__ X, unsigned rows_mask, int iters, float4* out) {
  const int lane = threadIdx.x & 63, g = lane >> 4, sub = lane & 15;
  unsigned state = (blockIdx.x * 4u + (threadIdx.x >> 6)) * 4u + g + 12345u;
  float4 acc = make_float4(0, 0, 0, 0);
  for (int it = 0; it < iters; ++it) {
    float4 x[NL];
#pragma unroll
    for (int t = 0; t < NL; ++t) {
      state = state * 1664525u + 1013904223u;
      const unsigned r = (state >> 8) & rows_mask;
      x[t] = X[(size_t)r * 16 + sub];
    }
#pragma unroll
    for (int t = 0; t < NL; ++t) { acc.x += x[t].x; acc.y += x[t].y; acc.z += x[t].z; acc.w += x[t].w; }
  }
  if (acc.x == 123.456f) out[0] = acc;
}

template <int NL>
void run(const float4* X, unsigned rows, int blocks_per_cu, float4* out) {
  const int blocks = 256 * blocks_per_cu, iters = 256 / NL * 4;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  gather<NL><<<blocks, 256>>>(X, rows - 1, iters, out);
  hipEventRecord(a);
  for (int r = 0; r < 5; ++r) gather<NL><<<blocks, 256>>>(X, rows - 1, iters, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double bytes = 5.0 * blocks * 16.0 * iters * NL * 256.0;   // 16 groups per block
  printf("rows %7u (%6.1f MB)  waves/CU %2d  in-flight/group %2d : %7.2f TB/s\n", rows, rows * 256.0 / 1e6,
         blocks_per_cu * 4, NL, bytes / (ms * 1e-3) / 1e12);
}

int main() {
  const size_t max_rows = 1u << 20;
  float4* X; float4* out;
  hipMalloc(&X, max_rows * 256); hipMalloc(&out, 64);
  hipMemset(X, 0, max_rows * 256);
  for (unsigned rows : {4096u, 16384u, 65536u, 262144u, 1048576u})
    for (int bpc : {2, 4, 8}) {
      run<4>(X, rows, bpc, out);
      run<8>(X, rows, bpc, out);
      run<16>(X, rows, bpc, out);
    }
  return 0;
}
