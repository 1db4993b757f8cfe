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
  // the index chunk of the NEXT iteration is in flight while this one's gathers are outstanding, so the
  // loop measures gather throughput, not the idx -> gather latency chain
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
    for (int t = 0; t < 8; ++t) x[t] = X[(size_t)cur[t] * stride4 + sub];
#pragma unroll
    for (int t = 0; t < 8; ++t) { acc.x += x[t].x; acc.y += x[t].y; acc.z += x[t].z; acc.w += x[t].w; }
  }
  if (acc.x == 123.456f) out[0] = acc;
}

// variant 5: does giving each XCD a FRACTION of the table pay?  Workgroup b runs on XCD b % 8; XCD x reads
// only the index stream of column class (x / (8 / K)) % K -- rows whose popularity rank is congruent to
// that class mod K -- so its L2 working set is 1/K of the table while the popularity law inside it is the
// same.  Same total number of 256-byte gathers as variant 0.
__global__ __launch_bounds__(256) void gather_split(const float4* __restrict__ X, const int* __restrict__ idx,
                                                    long per_class, int K, float4* out) {
  const int lane = threadIdx.x & 63, g = lane >> 4, sub = lane & 15;
  const int xcd = blockIdx.x & 7, cls = (xcd / (8 / K)) % K, peers = 8 / K;     // XCDs sharing this class
  const long wave_in_class = ((long)(blockIdx.x >> 3) * peers + (xcd % peers)) * 4 + (threadIdx.x >> 6);
  const long waves_in_class = (long)(gridDim.x >> 3) * peers * 4;
  const int* my = idx + (long)cls * per_class;
  float4 acc = make_float4(0, 0, 0, 0);
  for (long base = wave_in_class * 32; base + 32 <= per_class; base += waves_in_class * 32) {
    float4 x[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = X[(size_t)my[base + t * 4 + g] * 16 + sub];
#pragma unroll
    for (int t = 0; t < 8; ++t) { acc.x += x[t].x; acc.y += x[t].y; acc.z += x[t].z; acc.w += x[t].w; }
  }
  if (acc.x == 123.456f) out[0] = acc;
}

// variant 4: rows colder than a popularity rank are fetched with the non-temporal hint so they do not
// evict the popular rows from L2 (sign bit of the index = cold).
__global__ __launch_bounds__(256) void gather_nt(const float4* __restrict__ X, const int* __restrict__ enc, long n_idx,
                                                 float4* out) {
  const int lane = threadIdx.x & 63, g = lane >> 4, sub = lane & 15;
  const long wave = (blockIdx.x * 256L + threadIdx.x) >> 6, n_waves = gridDim.x * 4L;
  float4 acc = make_float4(0, 0, 0, 0);
  for (long base = wave * 32; base + 32 <= n_idx; base += n_waves * 32) {
    float4 x[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int e = enc[base + t * 4 + g];
      const float4* p = X + (size_t)(e & 0x7fffffff) * 16 + sub;
      typedef float v4f __attribute__((ext_vector_type(4)));
      if (e < 0) { const v4f q = __builtin_nontemporal_load((const v4f*)p); x[t] = make_float4(q.x, q.y, q.z, q.w); }
      else x[t] = *p;
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) { acc.x += x[t].x; acc.y += x[t].y; acc.z += x[t].z; acc.w += x[t].w; }
  }
  if (acc.x == 123.456f) out[0] = acc;
}

// variant 3: the H most popular rows are staged in LDS by every (persistent) workgroup; the index stream
// carries (0x80000000 | slot) for those.  Tests whether taking the hot lines off L2 pays.
__global__ __launch_bounds__(1024) void gather_hot(const float4* __restrict__ X, const int* __restrict__ enc,
                                                   const int* __restrict__ hot_rows, int H, long n_idx, float4* out) {
  extern __shared__ float4 hot[];
  for (int k = threadIdx.x; k < H * 16; k += 1024) hot[k] = X[(size_t)hot_rows[k >> 4] * 16 + (k & 15)];
  __syncthreads();
  const int lane = threadIdx.x & 63, g = lane >> 4, sub = lane & 15;
  const long wave = (blockIdx.x * 1024L + threadIdx.x) >> 6, n_waves = gridDim.x * 16L;
  float4 acc = make_float4(0, 0, 0, 0);
  for (long base = wave * 32; base + 32 <= n_idx; base += n_waves * 32) {
    float4 x[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int e = enc[base + t * 4 + g];
      x[t] = (e < 0) ? hot[(e & 0x7fffffff) * 16 + sub] : X[(size_t)e * 16 + sub];
    }
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
  std::vector<int> h(n_idx), rank(n_idx);
  std::uniform_real_distribution<double> U(0, s);
  for (long i = 0; i < n_idx; ++i) {
    rank[i] = (int)(std::lower_bound(cdf.begin(), cdf.end(), U(rng)) - cdf.begin());
    h[i] = perm[rank[i]];
  }
  float4 *X, *out; int* idx;
  CK(hipMalloc(&X, (size_t)rows * 256 * 2)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&idx, n_idx * 4));
  CK(hipMemset(X, 0, (size_t)rows * 256 * 2));
  CK(hipMemcpy(idx, h.data(), n_idx * 4, hipMemcpyHostToDevice));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int variant = 0; variant < 3; ++variant) {
    // 0: 256 B rows (stride 256)   1: 128 B half-rows inside 256 B rows (stride 256)   2: 128 B rows packed
    const int stride4 = (variant == 2) ? 8 : 16;
    for (int blocks : {1024, 2048, 4096}) {
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
  int *enc, *hot_rows;
  CK(hipMalloc(&enc, n_idx * 4)); CK(hipMalloc(&hot_rows, 1024 * 4));
  CK(hipMemcpy(hot_rows, perm.data(), 1024 * 4, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute((const void*)gather_hot, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int K : {1, 2, 4, 8}) {
    // class c = the fetches whose popularity rank is congruent to c mod K, padded to equal length
    std::vector<std::vector<int>> cls(K);
    for (long i = 0; i < n_idx; ++i) cls[rank[i] % K].push_back(h[i]);
    long per_class = 0;
    for (auto& c : cls) per_class = std::max<long>(per_class, (long)c.size());
    std::vector<int> flat((size_t)per_class * K);
    for (int c = 0; c < K; ++c)
      for (long i = 0; i < per_class; ++i) flat[(size_t)c * per_class + i] = cls[c][i % cls[c].size()];
    int* d_flat;
    CK(hipMalloc(&d_flat, flat.size() * 4));
    CK(hipMemcpy(d_flat, flat.data(), flat.size() * 4, hipMemcpyHostToDevice));
    for (int blocks : {2048, 4096}) {
      float ms = 0;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(a));
        gather_split<<<blocks, 256>>>(X, d_flat, per_class, K, out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms, a, b));
      }
      const double bytes = (double)per_class * K * 256;
      printf("variant 5 (256 B rows, each XCD reads 1/%d of the table = %.1f MB) blocks %4d: %7.2f us  %6.2f TB/s\n", K,
             rows * 256.0 / K / 1e6, blocks, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
    }
    CK(hipFree(d_flat));
  }
  for (int T : {38048}) {
    std::vector<int> e(n_idx);
    long cold = 0;
    for (long i = 0; i < n_idx; ++i) { const bool c = rank[i] >= T; cold += c; e[i] = c ? (int)(0x80000000u | h[i]) : h[i]; }
    CK(hipMemcpy(enc, e.data(), n_idx * 4, hipMemcpyHostToDevice));
    float ms = 0;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(a));
      gather_nt<<<2048, 256>>>(X, enc, n_idx, out);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      CK(hipEventElapsedTime(&ms, a, b));
    }
    printf("variant 4 (256 B rows, ranks >= %5d non-temporal = %4.1f%% of fetches): %7.2f us\n", T, 100.0 * cold / n_idx, ms * 1e3);
  }
  for (int H : {0, 256}) {
    std::vector<int> e(n_idx);
    long hits = 0;
    for (long i = 0; i < n_idx; ++i) { const bool hot = rank[i] < H; hits += hot; e[i] = hot ? (int)(0x80000000u | rank[i]) : h[i]; }
    CK(hipMemcpy(enc, e.data(), n_idx * 4, hipMemcpyHostToDevice));
    for (int blocks : {256, 512}) {
      if ((size_t)H * 256 * (blocks / 256) > 160 * 1024) continue;
      float ms = 0;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(a));
        gather_hot<<<blocks, 1024, (size_t)H * 256>>>(X, enc, hot_rows, H, n_idx, out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms, a, b));
      }
      printf("variant 3 (256 B rows, hot %3d rows in LDS = %4.1f%% of fetches) blocks %3d x 1024: %7.2f us\n", H,
             100.0 * hits / n_idx, blocks, ms * 1e3);
    }
  }
  return 0;
}
