// Premise test for "LDS-staged neighbour tiles" in the SpMM: do the gathers of the H most popular x rows get
// cheaper when every (persistent) workgroup keeps those rows in LDS?  Same fetch stream as gather_zipf.hip
// (2.52 M fetches of 256-byte rows, Zipf(0.8) popularity over 38,048 rows), 8 fetches in flight per 16-lane
// row-group, index stream software-pipelined.  What is compared, per H and per workgroups-per-CU:
//   base      every fetch from global memory (persistent 1024-thread workgroups, the H = 0 case of the others)
//   mixed     each LANE GROUP decides: hot -> ds_read_b128, cold -> global_load (the stream as the graph gives it)
//   round     the stream re-ordered so that the 4 row-groups of a wave agree in every round (uniform branch per load)
//   wave      ... and in all 8 rounds of an iteration (uniform branch per 8 loads)
//   halfmask  base with every second row-group predicated off in each load instruction (same instruction count,
//             half the bytes): does a load's cost follow its ACTIVE lanes?
// build: hipcc --offload-arch=gfx950 -O3 tools/microbench/gather_lds.hip -o tools/microbench/gather_lds.out
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

enum Mode { BASE = 0, MIXED = 1, ROUND = 2, WAVE = 3, HALFMASK = 4 };

template <int MODE>
__global__ __launch_bounds__(1024) void gather(const float4* __restrict__ X, const int* __restrict__ enc,
                                               const int* __restrict__ hot_rows, int H, long n_idx, float4* out) {
  extern __shared__ float4 hot[];
  for (int k = threadIdx.x; k < H * 16; k += 1024) hot[k] = X[(size_t)hot_rows[k >> 4] * 16 + (k & 15)];
  __syncthreads();
  const int lane = threadIdx.x & 63, g = lane >> 4, sub = lane & 15;
  const long wave = (blockIdx.x * 1024L + threadIdx.x) >> 6, n_waves = gridDim.x * 16L;
  float4 acc = make_float4(0, 0, 0, 0);
  long base = wave * 32;
  int nxt[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) nxt[t] = (base + 32 <= n_idx) ? enc[base + t * 4 + g] : 0;
  for (; base + 32 <= n_idx; base += n_waves * 32) {
    int cur[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) cur[t] = nxt[t];
    const long nb = base + n_waves * 32;
    if (nb + 32 <= n_idx) {
#pragma unroll
      for (int t = 0; t < 8; ++t) nxt[t] = enc[nb + t * 4 + g];
    }
    float4 x[8];
    if (MODE == BASE) {
#pragma unroll
      for (int t = 0; t < 8; ++t) x[t] = X[(size_t)cur[t] * 16 + sub];
    } else if (MODE == HALFMASK) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        x[t] = make_float4(0, 0, 0, 0);
        if ((t ^ g) & 1) x[t] = X[(size_t)cur[t] * 16 + sub];
      }
    } else if (MODE == MIXED) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int e = cur[t];
        x[t] = (e < 0) ? hot[(e & 0x7fffffff) * 16 + sub] : X[(size_t)e * 16 + sub];
      }
    } else if (MODE == ROUND) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int e = cur[t];
        if (__builtin_amdgcn_readfirstlane(e) < 0) x[t] = hot[(e & 0x7fffffff) * 16 + sub];
        else x[t] = X[(size_t)e * 16 + sub];
      }
    } else {
      if (__builtin_amdgcn_readfirstlane(cur[0]) < 0) {
#pragma unroll
        for (int t = 0; t < 8; ++t) x[t] = hot[(cur[t] & 0x7fffffff) * 16 + sub];
      } else {
#pragma unroll
        for (int t = 0; t < 8; ++t) x[t] = X[(size_t)cur[t] * 16 + sub];
      }
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) { acc.x += x[t].x; acc.y += x[t].y; acc.z += x[t].z; acc.w += x[t].w; }
  }
  if (acc.x == 123.456f) out[0] = acc;
}

// hot / cold fetches regrouped into pure blocks of `blk` entries, blocks shuffled (the tails are padded by repetition)
static std::vector<int> pure_blocks(const std::vector<int>& hot, const std::vector<int>& cold, int blk, long& n_out,
                                    std::mt19937_64& rng) {
  std::vector<std::vector<int>> blocks;
  auto cut = [&](const std::vector<int>& src) {
    for (size_t i = 0; i < src.size(); i += blk) {
      std::vector<int> b(blk);
      for (int k = 0; k < blk; ++k) b[k] = src[std::min(i + k, src.size() - 1)];
      blocks.push_back(std::move(b));
    }
  };
  cut(hot); cut(cold);
  std::shuffle(blocks.begin(), blocks.end(), rng);
  // an iteration is 32 entries laid out [round t][group g]; a block of 4 is one round, a block of 32 one iteration
  while ((blocks.size() * blk) % 32) blocks.push_back(blocks[0]);
  std::vector<int> flat;
  flat.reserve(blocks.size() * blk);
  for (auto& b : blocks) flat.insert(flat.end(), b.begin(), b.end());
  n_out = (long)flat.size();
  return flat;
}

int main() {
  const int rows = 38048;
  const long n_idx = 1260793L * 2;
  std::vector<double> cdf(rows);
  double s = 0;
  for (int k = 0; k < rows; ++k) { s += 1.0 / std::pow(k + 1.0, 0.8); cdf[k] = s; }
  std::mt19937_64 rng(1);
  std::vector<int> perm(rows);
  for (int k = 0; k < rows; ++k) perm[k] = k;
  std::shuffle(perm.begin(), perm.end(), rng);
  std::vector<int> h(n_idx), rank(n_idx);
  std::uniform_real_distribution<double> U(0, s);
  for (long i = 0; i < n_idx; ++i) {
    rank[i] = (int)(std::lower_bound(cdf.begin(), cdf.end(), U(rng)) - cdf.begin());
    h[i] = perm[rank[i]];
  }
  float4 *X, *out; int *enc, *hot_rows;
  CK(hipMalloc(&X, (size_t)rows * 256)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&enc, (n_idx + 4096) * 4));
  CK(hipMalloc(&hot_rows, 1024 * 4));
  CK(hipMemset(X, 0, (size_t)rows * 256));
  CK(hipMemcpy(hot_rows, perm.data(), 1024 * 4, hipMemcpyHostToDevice));      // perm[rank] = row of that rank
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const void* fns[5] = {(const void*)gather<BASE>, (const void*)gather<MIXED>, (const void*)gather<ROUND>,
                        (const void*)gather<WAVE>, (const void*)gather<HALFMASK>};
  for (const void* f : fns) CK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  auto time_it = [&](int mode, int blocks, int H, long n, float& us) -> int {
    float ms = 0, best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(a));
      const size_t lds = (size_t)H * 256;
      switch (mode) {
        case BASE: gather<BASE><<<blocks, 1024, lds>>>(X, enc, hot_rows, H, n, out); break;
        case MIXED: gather<MIXED><<<blocks, 1024, lds>>>(X, enc, hot_rows, H, n, out); break;
        case ROUND: gather<ROUND><<<blocks, 1024, lds>>>(X, enc, hot_rows, H, n, out); break;
        case WAVE: gather<WAVE><<<blocks, 1024, lds>>>(X, enc, hot_rows, H, n, out); break;
        default: gather<HALFMASK><<<blocks, 1024, lds>>>(X, enc, hot_rows, H, n, out); break;
      }
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      CK(hipEventElapsedTime(&ms, a, b));
      if (rep) best = std::min(best, ms);
    }
    us = best * 1e3f;
    return 0;
  };
  float us;
  // ---- all fetches from global memory; every second row-group masked
  CK(hipMemcpy(enc, h.data(), n_idx * 4, hipMemcpyHostToDevice));
  for (int blocks : {256, 512}) {
    if (time_it(BASE, blocks, 0, n_idx, us)) return 1;
    printf("base      H   0  %d x 1024 threads: %7.2f us  (%5.2f TB/s)\n", blocks, us, n_idx * 256.0 / us / 1e6);
    if (time_it(HALFMASK, blocks, 0, n_idx, us)) return 1;
    printf("halfmask  H   0  %d x 1024 threads: %7.2f us  (half the row-groups of every load predicated off)\n", blocks, us);
  }
  for (int H : {128, 256, 512}) {
    std::vector<int> e(n_idx), hot, cold;
    for (long i = 0; i < n_idx; ++i) {
      const bool is_hot = rank[i] < H;
      e[i] = is_hot ? (int)(0x80000000u | (unsigned)rank[i]) : h[i];
      (is_hot ? hot : cold).push_back(e[i]);
    }
    const double frac = 100.0 * hot.size() / n_idx;
    for (int blocks : {256, 512}) {
      if ((size_t)H * 256 * (blocks / 256) > 150 * 1024) continue;
      CK(hipMemcpy(enc, e.data(), n_idx * 4, hipMemcpyHostToDevice));
      if (time_it(MIXED, blocks, H, n_idx, us)) return 1;
      printf("mixed     H %3d (%4.1f %% of the fetches)  %d x 1024: %7.2f us\n", H, frac, blocks, us);
      for (int blk : {4, 32}) {
        long n2;
        std::vector<int> flat = pure_blocks(hot, cold, blk, n2, rng);
        if (n2 > n_idx + 4096) { printf("stream too long\n"); return 1; }
        CK(hipMemcpy(enc, flat.data(), n2 * 4, hipMemcpyHostToDevice));
        if (time_it(blk == 4 ? ROUND : WAVE, blocks, H, n2, us)) return 1;
        printf("%s H %3d (%4.1f %% of the fetches)  %d x 1024: %7.2f us\n", blk == 4 ? "round    " : "wave     ", H, frac, blocks, us);
      }
    }
  }
  return 0;
}
