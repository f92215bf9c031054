// Microbenchmark: how long does ONE vector-memory instruction hold its wave's issue?  A single wave of a workgroup issues
// N loads back to back between two s_memtime stamps (N = 1, 2, 4, 8; the slope is the per-instruction cost, the intercept
// the stamps' own), the other waves idle (s_sleep) or run MFMAs:
//   kind 0: global_load_lds_dwordx4 (LDS-DMA, 1 KiB per wave-instruction, M0 = LDS base)
//   kind 1: global_load_dwordx4 to VGPRs (asynchronous until its s_waitcnt)
//   kind 2: global_load_lds_dword (LDS-DMA, 256 B per wave-instruction)
//   kind 3: global_load_dwordx4 to VGPRs + ds_write_b128 (register staging, the writes behind the loads' vmcnt)
// Sources: a 256 KB buffer every workgroup shares (L2 hits after the first touch), 1 workgroup per CU, 256 workgroups.
//   hipcc --offload-arch=gfx950 -O3 -o vmem_issue_cost vmem_issue_cost.hip && ./vmem_issue_cost
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define GLDS(gptr, lptr, bytes)                                                                \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),      \
                                   (__attribute__((address_space(3))) void*)(lptr), bytes, 0, 0)

template <int KIND, int N, bool MFMA, int PRIO = 0>
__global__ __launch_bounds__(512, 2) void k(const u32x4* src, float* out, unsigned long long* stamps, int rounds) {
  __shared__ __attribute__((aligned(16))) u32x4 lds16[4096];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  float r = 0.f;
  if (wv == 0) {
    unsigned long long tot = 0;
    u32x4 v[8];
    if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
    for (int it = 0; it < rounds; ++it) {
      const u32x4* p = src + (long)((it * 8) % 64) * 256 + lane;
      unsigned long long s0, s1;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s0)::"memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < N; ++i) {
        if (KIND == 0) GLDS(p + i * 64, lds16 + i * 64, 16);
        else if (KIND == 2) GLDS((const unsigned*)(p + i * 64) + lane * 0, (unsigned*)lds16 + i * 64, 4);
        else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[i]) : "v"(p + i * 64) : "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s1)::"memory");
      __builtin_amdgcn_sched_barrier(0);
      tot += s1 - s0;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (KIND == 1 || KIND == 3) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
          asm volatile("" : "+v"(v[i]));
          if (KIND == 3) lds16[i * 64 + lane] = v[i];
          else r += __uint_as_float(v[i][0] & 0x3fffffff);
        }
      }
      __builtin_amdgcn_s_sleep(20);
    }
    if (lane == 0) stamps[blockIdx.x] = tot / rounds;
  } else if (MFMA) {
    bf16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.01f * (float)((tid + j) % 13)); b[j] = (__bf16)(0.02f * (float)((tid * 3 + j) % 11)); }
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
    for (int it = 0; it < rounds * 12; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < 16; ++q) r += acc[i][q];
  }
  __syncthreads();
  if (KIND == 0 || KIND == 2 || KIND == 3) r += __uint_as_float(lds16[tid][0] & 0x3fffffff);
  out[(long)blockIdx.x * 512 + tid] = r;
}

template <int KIND, int N, bool MFMA, int PRIO = 0>
static double run(const u32x4* src, float* out, unsigned long long* st) {
  const int nwg = 256;
  hipLaunchKernelGGL((k<KIND, N, MFMA, PRIO>), dim3(nwg), dim3(512), 0, 0, src, out, st, 200);
  hipLaunchKernelGGL((k<KIND, N, MFMA, PRIO>), dim3(nwg), dim3(512), 0, 0, src, out, st, 200);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(nwg);
  (void)hipMemcpy(h.data(), st, nwg * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  return (double)h[nwg / 2];
}

template <int KIND, bool MFMA, int PRIO = 0>
static void row(const char* name, const u32x4* src, float* out, unsigned long long* st) {
  const double c1 = run<KIND, 1, MFMA, PRIO>(src, out, st), c2 = run<KIND, 2, MFMA, PRIO>(src, out, st), c4 = run<KIND, 4, MFMA, PRIO>(src, out, st),
               c8 = run<KIND, 8, MFMA, PRIO>(src, out, st);
  printf("%-58s %s | cycles between the stamps: N=1 %5.0f  N=2 %5.0f  N=4 %5.0f  N=8 %5.0f | per instruction (N 8 vs 1) %5.1f\n", name,
         MFMA ? "beside 7 MFMA waves" : "other waves idle   ", c1, c2, c4, c8, (c8 - c1) / 7.0);
  fflush(stdout);
}

int main() {
  u32x4* src;
  float* out;
  unsigned long long* st;
  (void)hipMalloc(&src, 64L * 256 * 16 * 2);
  (void)hipMalloc(&out, 256L * 512 * 4);
  (void)hipMalloc(&st, 256 * 8);
  (void)hipMemset(src, 0x11, 64L * 256 * 16 * 2);
  row<0, false>("global_load_lds_dwordx4 (LDS-DMA 1 KiB)", src, out, st);
  row<0, true>("global_load_lds_dwordx4 (LDS-DMA 1 KiB)", src, out, st);
  row<2, false>("global_load_lds_dword (LDS-DMA 256 B)", src, out, st);
  row<2, true>("global_load_lds_dword (LDS-DMA 256 B)", src, out, st);
  row<1, false>("global_load_dwordx4 -> VGPR", src, out, st);
  row<1, true>("global_load_dwordx4 -> VGPR", src, out, st);
  row<0, true, 1>("LDS-DMA 1 KiB, issuing wave at s_setprio 1", src, out, st);
  row<0, true, 3>("LDS-DMA 1 KiB, issuing wave at s_setprio 3", src, out, st);
  return 0;
}
