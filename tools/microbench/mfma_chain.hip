// Which scalar fma order reproduces v_mfma_f32_32x32x2_f32's accumulation bit for bit?  (csrc/eval.hip re-scores the
// survivors of its split-bf16 filter pass; a VALU chain would cost 1/32 of replicating one user over an MFMA tile, but
// it may only replace the MFMA if it yields the SAME BITS as gemm_nt_kernel.)  gemm_nt_kernel's operand assignment:
// lane half h holds dimensions [h*DH, (h+1)*DH); MFMA step s multiplies k = s (half 0) and k = DH + s (half 1).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_chain.hip -o /tmp/mfma_chain && /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
constexpr int D = 64, DH = 32;

__global__ void mfma_tile(const float* A, const float* B, float* C) {   // A, B: 32 x D row-major; C: 32 x 32
  const int lane = threadIdx.x, r32 = lane & 31, h = lane >> 5;
  float a[DH], b[DH];
  for (int t = 0; t < DH; ++t) { a[t] = A[r32 * D + h * DH + t]; b[t] = B[r32 * D + h * DH + t]; }
  floatx16 acc;
  for (int t = 0; t < 16; ++t) acc[t] = 0.f;
#pragma unroll
  for (int s = 0; s < DH; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
  for (int t = 0; t < 16; ++t) C[((t & 3) + 8 * (t >> 2) + 4 * h) * 32 + r32] = acc[t];
}

// variant 0: per step s: acc = fma(a[s], b[s], acc); acc = fma(a[DH+s], b[DH+s], acc)
// variant 1: per step s: the two k's in the other order
// variant 2: per step s: acc = acc + fma(a[DH+s], b[DH+s], a[s]*b[s])   (pair summed first, products exact-ish)
// variant 3: per step s: t = fma(a[s], b[s], acc) then acc = fma(a1, b1, t) but with non-fused multiply-add (mul, add)
// variant 4: plain k = 0 .. D-1 sequential fma
__global__ void valu_tile(const float* A, const float* B, float* C, int variant) {
  const int i = threadIdx.x >> 5, j = threadIdx.x & 31;                 // 1024 threads: one output each
  const float* a = A + i * D;
  const float* b = B + j * D;
  float acc = 0.f;
  if (variant == 4) {
    for (int k = 0; k < D; ++k) acc = __builtin_fmaf(a[k], b[k], acc);
  } else {
    for (int s = 0; s < DH; ++s) {
      const float a0 = a[s], b0 = b[s], a1 = a[DH + s], b1 = b[DH + s];
      if (variant == 0) { acc = __builtin_fmaf(a0, b0, acc); acc = __builtin_fmaf(a1, b1, acc); }
      else if (variant == 1) { acc = __builtin_fmaf(a1, b1, acc); acc = __builtin_fmaf(a0, b0, acc); }
      else if (variant == 2) { acc = acc + __builtin_fmaf(a1, b1, a0 * b0); }
      else { acc = __fadd_rn(acc, __fmul_rn(a0, b0)); acc = __fadd_rn(acc, __fmul_rn(a1, b1)); }
    }
  }
  C[i * 32 + j] = acc;
}

int main() {
  float *dA, *dB, *dC, *dV;
  hipMalloc(&dA, 32 * D * 4); hipMalloc(&dB, 32 * D * 4); hipMalloc(&dC, 1024 * 4); hipMalloc(&dV, 1024 * 4);
  std::vector<float> A(32 * D), B(32 * D), C(1024), V(1024);
  int match[5] = {0, 0, 0, 0, 0}, trials = 0;
  for (int trial = 0; trial < 40; ++trial) {
    srand(trial);
    const float scale = (trial % 4 == 0) ? 1.f : (trial % 4 == 1) ? 1e-3f : (trial % 4 == 2) ? 300.f : 0.1f;
    for (auto& x : A) x = scale * ((float)rand() / RAND_MAX - 0.5f);
    for (auto& x : B) x = scale * ((float)rand() / RAND_MAX - 0.5f);
    if (trial % 5 == 4) for (int k = 0; k < D; k += 2) { B[k + 1] = -B[k]; A[k + 1] = A[k] * (1.f + 1e-6f * (k + 1)); }   // cancellation
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    mfma_tile<<<1, 64>>>(dA, dB, dC);
    hipMemcpy(C.data(), dC, 1024 * 4, hipMemcpyDeviceToHost);
    for (int v = 0; v < 5; ++v) {
      valu_tile<<<1, 1024>>>(dA, dB, dV, v);
      hipMemcpy(V.data(), dV, 1024 * 4, hipMemcpyDeviceToHost);
      match[v] += memcmp(C.data(), V.data(), 1024 * 4) == 0;
    }
    ++trials;
  }
  const char* names[5] = {"fma k=s then k=DH+s", "fma k=DH+s then k=s", "acc + fma(a1,b1,a0*b0)", "unfused mul, add", "fma k = 0..D-1"};
  for (int v = 0; v < 5; ++v) printf("variant %d (%s): %d / %d tiles bit-identical to v_mfma_f32_32x32x2_f32\n", v, names[v], match[v], trials);
  return 0;
}
