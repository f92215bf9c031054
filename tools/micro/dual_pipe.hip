// Microbenchmark: can a CU sustain fp32 MFMA waves and fp32 VALU-FMA waves at the same time?
// mode 0: 4 MFMA waves/CU; mode 1: 4 VALU waves/CU; mode 2: 4 MFMA + 4 VALU waves/CU; mode 3: 8 VALU waves/CU;
// mode 4: 4 MFMA + 8 VALU waves per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(768) void k(float* out, int iters, int mode, int nmf) {
  const int wv = threadIdx.x >> 6;
  const bool mfma = (mode == 0) || ((mode == 2 || mode == 4) && wv < nmf);
  float r = 0.f;
  if (mfma) {
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = threadIdx.x * 1e-3f + 1.0f, y = 0.5f + blockIdx.x * 1e-4f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
      }
    }
    r = a0[0] + a1[1] + a2[2] + a3[3];
  } else {
    float acc[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) acc[j] = j * 0.01f;
    float x = threadIdx.x * 1e-3f + 1.0f, y = 1.0f - blockIdx.x * 1e-6f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
#pragma unroll
        for (int j = 0; j < 32; ++j) acc[j] = __builtin_fmaf(acc[j], y, x);
      }
    }
#pragma unroll
    for (int j = 0; j < 32; ++j) r += acc[j];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 8 * 768 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  struct { int mode, threads, nmf; const char* name; } cfg[] = {
      {0, 256, 4, "4 MFMA waves/CU"}, {1, 256, 0, "4 VALU waves/CU"}, {3, 512, 0, "8 VALU waves/CU"},
      {2, 512, 4, "4 MFMA + 4 VALU waves/CU"}, {4, 768, 4, "4 MFMA + 8 VALU waves/CU"}};
  for (auto& c : cfg) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(256 * 4), dim3(c.threads), 0, 0, d, iters, c.mode, c.nmf);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const int waves = c.threads / 64;
    const int nm = (c.mode == 0) ? waves : ((c.mode == 2 || c.mode == 4) ? c.nmf : 0);
    const int nv = waves - nm;
    const double fm = 1024.0 * nm * (double)iters * 16 * 4096.0;           // 16 MFMAs/iter x 4096 FLOP
    const double fv = 1024.0 * nv * (double)iters * 16 * 32 * 64 * 2.0;    // 512 FMA instr/iter x 64 lanes x 2
    printf("%-28s %8.3f ms  MFMA %7.1f TF  VALU %7.1f TF  total %7.1f TF\n", c.name, ms, fm / ms / 1e9, fv / ms / 1e9,
           (fm + fv) / ms / 1e9);
  }
  return 0;
}
