// Does halving the per-XCD gather working set pay?  Gathers of ROWB-byte rows (256 or 128) addressed by
// a precomputed Zipf(0.8)-distributed index stream (the item-popularity law of the synthetic graph),
// 8 in flight per row-group, table of `rows` rows.  Prints effective gather bandwidth.
// build: hipcc --offload-arch=gfx950 -O3 tools/microbench/gather_zipf.hip -o /tmp/gather_zipf
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s\n", hipGetErrorString(e_)); return 1; } } while (0)

template <int LPR>   // lanes per row: 16 -> 256 B rows, 8 -> 128 B rows
__global__ __launch_bounds__(256) void gather(const float4* __restrict__ X, const int* __restrict__ idx, long n_idx,
                                              int stride4, float4* out) {
  constexpr int G = 64 / LPR;
  const int lane = threadIdx.x & 63, g = lane / LPR, sub = lane % LPR;
  const long wave = (blockIdx.x * 256L + threadIdx.x) >> 6, n_waves = gridDim.x * 4L;
  float4 acc = make_float4(0, 0, 0, 0);
  for (long base = wave * 8 * G; base + 8 * G <= n_idx; base += n_waves * 8 * G) {
    float4 x[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = X[(size_t)idx[base + t * G + g] * stride4 + sub];
#pragma unroll
    for (int t = 0; t < 8; ++t) { acc.x += x[t].x; acc.y += x[t].y; acc.z += x[t].z; acc.w += x[t].w; }
  }
  if (acc.x == 123.456f) out[0] = acc;
}

int main() {
  const int rows = 38048;
  const long n_idx = 1260793L * 2;     // as many row fetches as one SpMM does per table class x2
  std::vector<double> cdf(rows);
  double s = 0;
  for (int k = 0; k < rows; ++k) { s += 1.0 / std::pow(k + 1.0, 0.8); cdf[k] = s; }
  std::mt19937_64 rng(1);
  std::vector<int> perm(rows);
  for (int k = 0; k < rows; ++k) perm[k] = k;
  std::shuffle(perm.begin(), perm.end(), rng);
  std::vector<int> h(n_idx);
  std::uniform_real_distribution<double> U(0, s);
  for (long i = 0; i < n_idx; ++i) h[i] = perm[std::lower_bound(cdf.begin(), cdf.end(), U(rng)) - cdf.begin()];
  float4 *X, *out; int* idx;
  CK(hipMalloc(&X, (size_t)rows * 256 * 2)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&idx, n_idx * 4));
  CK(hipMemset(X, 0, (size_t)rows * 256 * 2));
  CK(hipMemcpy(idx, h.data(), n_idx * 4, hipMemcpyHostToDevice));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int variant = 0; variant < 3; ++variant) {
    // 0: 256 B rows (stride 256)   1: 128 B half-rows inside 256 B rows (stride 256)   2: 128 B rows packed
    const int stride4 = (variant == 2) ? 8 : 16;
    for (int blocks : {2048, 4096}) {
      float ms = 0;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(a));
        if (variant == 0) gather<16><<<blocks, 256>>>(X, idx, n_idx, stride4, out);
        else gather<8><<<blocks, 256>>>(X, idx, n_idx, stride4, out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms, a, b));
      }
      const double bytes = (double)n_idx * (variant == 0 ? 256 : 128);
      printf("variant %d (%s) blocks %4d: %7.2f us  %6.2f TB/s   working set %.1f MB\n", variant,
             variant == 0 ? "256 B rows" : variant == 1 ? "128 B half of 256 B rows" : "128 B rows packed", blocks,
             ms * 1e3, bytes / (ms * 1e-3) / 1e12, rows * (variant == 0 ? 256.0 : 128.0) / 1e6);
    }
  }
  return 0;
}
