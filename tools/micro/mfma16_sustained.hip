// Microbenchmark: what does the matrix pipe SUSTAIN on dense 16-bit MFMA loops with random operands?
// (MI355X_MICROARCH.md "DVFS give-back": under dense bf16/f16 MFMA load the chip lowers its clock, and the clock it
// holds depends on the data and on the MFMA shape; the 2.5 PFLOP/s "peak" assumes 2.4 GHz.)
// Bare loops, operands in registers, the wave tile of conv27_bf16 (64 couts x 128 voxels per wave):
//   shape 0: v_mfma_f32_32x32x16  2 A fragments x 4 B fragments =  8 MFMAs / k-step of 16
//   shape 1: v_mfma_f32_16x16x32  4 A fragments x 8 B fragments = 32 MFMAs / k-step of 32
// x {bf16, f16} x {random operands, zeros} x {1, 2 waves per SIMD}.  Every configuration runs back to back for ~2 s; the
// in-kernel clock is (delta s_memtime / delta s_memrealtime) x 100 MHz, median over workgroups (stamps go to their own
// buffer, no output depends on them).
//   hipcc --offload-arch=gfx950 -O3 -o mfma16_sustained mfma16_sustained.hip && ./mfma16_sustained
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float urand(unsigned s) { return (float)(hash32(s) >> 8) * (2.0f / 16777216.0f) - 1.0f; }   // [-1, 1)

template <typename V, typename E>
__device__ __forceinline__ V make_frag(unsigned seed, int zero) {
  V v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = zero ? (E)0.0f : (E)urand(seed * 8u + j);
  return v;
}

template <int SHAPE, bool F16>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* stamps, int iters, int zero) {
  using V = typename std::conditional<F16, f16x8, bf16x8>::type;
  using E = typename std::conditional<F16, _Float16, __bf16>::type;
  const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
  float r = 0.f;
  unsigned long long t0 = 0, t1 = 0, r0 = 0, r1 = 0;
  if (SHAPE == 0) {
    V a[2], b[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i] = make_frag<V, E>(gid * 16u + i, zero);
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = make_frag<V, E>(gid * 16u + 8 + i, zero);
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (F16) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16((f16x8)a[i], (f16x8)b[j], acc[i][j], 0, 0, 0);
            else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((bf16x8)a[i], (bf16x8)b[j], acc[i][j], 0, 0, 0);
          }
    }
    t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) r += acc[i][j][(i + j) & 15];
  } else {
    V a[4], b[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = make_frag<V, E>(gid * 16u + i, zero);
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = make_frag<V, E>(gid * 16u + 8 + i, zero);
    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[i][j][q] = 0.f;
    t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (F16) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16((f16x8)a[i], (f16x8)b[j], acc[i][j], 0, 0, 0);
            else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((bf16x8)a[i], (bf16x8)b[j], acc[i][j], 0, 0, 0);
          }
    }
    t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) r += acc[i][j][(i + j) & 3];
  }
  out[gid] = r;
  if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int SHAPE, bool F16>
static void run(const char* name, float* d, unsigned long long* ds, int threads, int zero) {
  const int blocks = 256, iters = 4000;
  // FLOP per iteration per wave: shape 0: 4 x 8 MFMAs x 2*32*32*16; shape 1: 2 x 32 MFMAs x 2*16*16*32 -- identical
  const double flop_per_launch = (double)blocks * (threads / 64) * iters * 32.0 * 32768.0;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<SHAPE, F16>), dim3(blocks), dim3(threads), 0, 0, d, ds, iters, zero);
  hipDeviceSynchronize();
  // ~2 s back to back, then time the last batch of launches
  float ms1 = 0.f;
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<SHAPE, F16>), dim3(blocks), dim3(threads), 0, 0, d, ds, iters, zero);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  hipEventElapsedTime(&ms1, e0, e1);
  const int reps = std::max(8, (int)(2000.0f / ms1));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<SHAPE, F16>), dim3(blocks), dim3(threads), 0, 0, d, ds, iters, zero);
  const int timed = 8;
  hipEventRecord(e0);
  for (int i = 0; i < timed; ++i) hipLaunchKernelGGL((k<SHAPE, F16>), dim3(blocks), dim3(threads), 0, 0, d, ds, iters, zero);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= timed;
  std::vector<unsigned long long> st(blocks * 2);
  hipMemcpy(st.data(), ds, st.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> clk;
  for (int b = 0; b < blocks; ++b)
    if (st[b * 2 + 1]) clk.push_back((double)st[b * 2] / (double)st[b * 2 + 1] * 0.1);      // GHz (s_memrealtime ticks at 100 MHz)
  std::sort(clk.begin(), clk.end());
  const double ghz = clk.empty() ? 0.0 : clk[clk.size() / 2];
  const double tf = flop_per_launch / (ms * 1e-3) / 1e12;
  const double cyc_per_mfma = ghz * 1e9 * (ms * 1e-3) / ((double)iters * 32.0 * (threads / 256) ) / (SHAPE == 0 ? 1.0 : 1.0);
  printf("%-34s %d wave/SIMD %-6s %8.3f ms  %7.1f TFLOP/s = %5.3f of 2500   clock %5.3f GHz  (%5.1f cyc per 32x32x16-equivalent)\n",
         name, threads / 256, zero ? "zeros" : "random", ms, tf, tf / 2500.0, ghz, cyc_per_mfma);
  hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
  float* d; unsigned long long* ds;
  hipMalloc(&d, 256 * 512 * sizeof(float));
  hipMalloc(&ds, 256 * 2 * sizeof(unsigned long long));
  for (int threads : {256, 512})
    for (int zero : {0, 1}) {
      run<0, false>("v_mfma_f32_32x32x16_bf16", d, ds, threads, zero);
      run<1, false>("v_mfma_f32_16x16x32_bf16", d, ds, threads, zero);
      run<0, true>("v_mfma_f32_32x32x16_f16", d, ds, threads, zero);
      run<1, true>("v_mfma_f32_16x16x32_f16", d, ds, threads, zero);
    }
  return 0;
}
