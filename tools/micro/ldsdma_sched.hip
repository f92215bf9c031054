// Microbenchmark: what does LDS-DMA staging cost an MFMA-paced 8-wave workgroup, by WHERE the pieces are issued and HOW
// the waves synchronise?  A model of conv27_bf16's K loop without its LDS reads: per stage a wave issues 72
// v_mfma_f32_32x32x16_bf16 (9 "taps" of 8, operands in registers) and its share of the stage's LDS-DMA pieces
// (1 KiB each: global_load_lds_dwordx4), weights from a slab every workgroup shares (L2-hot) and activations from a
// region of the workgroup's own (streamed); two LDS buffers as in the kernel.
//   sync  0: free running (no barrier, vmcnt(0) once per stage)   1: vmcnt(0) + s_barrier per stage (the lockstep kernel)
//   sched[w][tap]: pieces wave w issues behind the MFMAs of `tap` (a 72-digit string per configuration)
// Reports cycles per stage (MFMA-ideal: 4608 at two waves per SIMD) and the cycles a wave spends blocked issuing pieces.
//   hipcc --offload-arch=gfx950 -O3 -o ldsdma_sched ldsdma_sched.hip && ./ldsdma_sched
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define GLDS16(gptr, lptr)                                                                     \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),      \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

struct Sched { unsigned long long n[8]; };      // 4 bits per tap

constexpr int BUF_BYTES = 57344;          // one stage buffer (weights 36 KB + halo 20 KB), two of them

template <int SYNC, int READS>
__global__ __launch_bounds__(512, 2) void k(const unsigned short* wsl, const unsigned short* xsl, float* out,
                                            unsigned long long* stamps, int nstage, long x_wg_stride, Sched sc, int wpieces, int mfma_mask, int wslabs, int xshared, int posmode) {
  extern __shared__ __attribute__((aligned(16))) u32x4 lds16[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  bf16x8 a[2], b[4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) a[i][j] = (__bf16)(0.001f * (float)((tid * 7 + i * 3 + j) % 97) - 0.05f);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) b[i][j] = (__bf16)(0.002f * (float)((tid * 5 + i * 11 + j) % 89) - 0.09f);
  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  // piece p of wave wv in stage st: p < wpieces: weights (shared slab, advancing with the stage); else this workgroup's halo rows
  const unsigned short* xw = xsl + (long)(xshared ? (blockIdx.x & 7) : blockIdx.x) * x_wg_stride;
  const bool do_mfma = (mfma_mask >> wv) & 1;
  // position inside a tap's 8 MFMAs behind which this wave issues its pieces: 7 = behind the whole tap (all waves together);
  // posmode 1: wave w behind its MFMA w (partners w, w + 4 sit on one SIMD and alternate MFMAs: the eight waves' slots are ~64 cycles apart)
  const int mypos = posmode == 1 ? wv : 7;
  unsigned long long t_blocked = 0, t0, t1;
  const unsigned long long mysched = __builtin_amdgcn_readfirstlane((unsigned)(sc.n[wv] & 0xffffffffull)) |
                                     ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(sc.n[wv] >> 32)) << 32);
  __syncthreads();
  t0 = __builtin_amdgcn_s_memtime();
  if (do_mfma) {
  for (int st = 0; st < nstage; ++st) {
    u32x4* base = lds16 + ((st + 1) & 1) * (BUF_BYTES / 16);
    int pidx = 0;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (READS) {
        // the kernel's fragment reads: 6 ds_read_b128 per tap from the current buffer
        const u32x4* cur = lds16 + (st & 1) * (BUF_BYTES / 16);
        __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = __builtin_bit_cast(bf16x8, cur[tap * 256 + i * 128 + lane * 2]);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = __builtin_bit_cast(bf16x8, cur[2304 + (tap * 3 + j * 100 + wv * 64 + lane) % 1200]);
      }
      __builtin_amdgcn_sched_barrier(0);
      const int n = (int)((mysched >> (4 * tap)) & 15ull);
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        acc[m >> 2][m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m >> 2], b[m & 3], acc[m >> 2][m & 3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      if (n && m == mypos) {
        unsigned long long s0, s1;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s0)::"memory");
        for (int i = 0; i < n; ++i, ++pidx) {
          const unsigned short* src = pidx < wpieces ? wsl + ((long)(st % wslabs) * 4608 + (long)(pidx * 8 + wv) * 64 + lane) * 8
                                                     : xw + ((long)(st % 16) * 2560 + (long)((((pidx - wpieces) & 1) * 8 + wv) * 64 + lane)) * 8;
          GLDS16(src, base + ((pidx * 8 + wv) * 64) % 3520);       // stays inside the 3584-slot buffer
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s1)::"memory");
        t_blocked += s1 - s0;
      }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (SYNC == 1) __builtin_amdgcn_s_barrier();
  }
  } else {
  for (int st = 0; st < nstage; ++st) {
    u32x4* base = lds16 + ((st + 1) & 1) * (BUF_BYTES / 16);
    int pidx = 0;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (READS) {
        // the kernel's fragment reads: 6 ds_read_b128 per tap from the current buffer
        const u32x4* cur = lds16 + (st & 1) * (BUF_BYTES / 16);
        __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = __builtin_bit_cast(bf16x8, cur[tap * 256 + i * 128 + lane * 2]);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = __builtin_bit_cast(bf16x8, cur[2304 + (tap * 3 + j * 100 + wv * 64 + lane) % 1200]);
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_sleep(2);
      __builtin_amdgcn_sched_barrier(0);
      const int n = (int)((mysched >> (4 * tap)) & 15ull);
      if (n) {
        unsigned long long s0, s1;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s0)::"memory");
        for (int i = 0; i < n; ++i, ++pidx) {
          const unsigned short* src = pidx < wpieces ? wsl + ((long)(st % wslabs) * 4608 + (long)(pidx * 8 + wv) * 64 + lane) * 8
                                                     : xw + ((long)(st % 16) * 2560 + (long)((((pidx - wpieces) & 1) * 8 + wv) * 64 + lane)) * 8;
          GLDS16(src, base + ((pidx * 8 + wv) * 64) % 3520);       // stays inside the 3584-slot buffer
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s1)::"memory");
        t_blocked += s1 - s0;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (SYNC == 1) __builtin_amdgcn_s_barrier();
  }
  }
  t1 = __builtin_amdgcn_s_memtime();
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) r += acc[i][j][q];
  out[(long)blockIdx.x * 512 + tid] = r;
  if (lane == 0) {
    stamps[((long)blockIdx.x * 8 + wv) * 2] = t1 - t0;
    stamps[((long)blockIdx.x * 8 + wv) * 2 + 1] = t_blocked;
  }
}

static Sched parse(const char* s72) {
  Sched sc;
  memset(&sc, 0, sizeof(sc));
  for (int w = 0; w < 8; ++w)
    for (int t = 0; t < 9; ++t) sc.n[w] |= (unsigned long long)(s72[w * 9 + t] - '0') << (4 * t);
  return sc;
}

int main() {
  const int nwg = 1024, nstage = 48;
  unsigned short *wsl, *xsl;
  float* out;
  unsigned long long* st;
  const long x_wg_stride = 16L * 2560 * 8;                                   // 16 distinct halo tiles per workgroup (655 KB)
  hipMalloc(&wsl, 64L * 4608 * 8 * 2);                                       // 64 stage slabs of 36 KB (4.7 MB: L2 / MALL resident)
  hipMalloc(&xsl, (long)nwg * x_wg_stride * 2);                              // 671 MB: streamed
  hipMalloc(&out, (long)nwg * 512 * 4);
  hipMalloc(&st, (long)nwg * 16 * 8);
  hipMemset(wsl, 0x11, 64L * 4608 * 8 * 2);
  hipMemset(xsl, 0x22, (long)nwg * x_wg_stride * 2);
  struct Cfg { const char* name; const char* s; int mfma_mask; int wslabs; int xshared; int sync; int reads; int posmode; };
  const char* NONE = "000000000" "000000000" "000000000" "000000000" "000000000" "000000000" "000000000" "000000000";
  const char* BURST = "700000000" "700000000" "700000000" "700000000" "700000000" "700000000" "700000000" "700000000";
  const char* SPREAD = "111111100" "111111100" "111111100" "111111100" "111111100" "111111100" "111111100" "111111100";
  const char* LOADER4 = "222222200" "222222200" "222222200" "222222200" "000000000" "000000000" "000000000" "000000000";   // waves 0-3 load everything (14 each)
  const char* SPREAD8 = "211111100" "211111100" "211111100" "211111100" "211111100" "211111100" "211111100" "211111100";
  const char* HALF = "000000000" "000000000" "000000000" "000000000" "700000000" "700000000" "700000000" "700000000";
  const char* HALFS = "000000000" "000000000" "000000000" "000000000" "111111100" "111111100" "111111100" "111111100";
  const char* ONE = "000000000" "000000000" "000000000" "000000000" "111111100" "000000000" "000000000" "000000000";
  std::vector<Cfg> cfgs = {
      {"MFMA on waves 0-3 only (one per SIMD), no DMA", NONE, 0x0f, 8, 0, 1, 0, 0},
      {"MFMA w0-3; w4-7 sleep + burst 7 pieces behind tap 0", HALF, 0x0f, 8, 0, 1, 0, 0},
      {"MFMA w0-3; w4-7 sleep + 1 piece per tap", HALFS, 0x0f, 8, 0, 1, 0, 0},
      {"MFMA w0-3; only wave 4 issues 1 piece per tap", ONE, 0x0f, 8, 0, 1, 0, 0},
      {"MFMA all 8; only wave 4 issues 1 piece per tap", ONE, 0xff, 8, 0, 1, 0, 0},
      {"MFMA all 8, no DMA", NONE, 0xff, 8, 0, 1, 0, 0},
      {"nobody computes; w4-7 burst 7 pieces", HALF, 0x00, 8, 0, 1, 0, 0},
  };
  for (auto& c : cfgs) {
    Sched sc = parse(c.s);
    auto kern = c.sync ? (c.reads ? k<1, 1> : k<1, 0>) : (c.reads ? k<0, 1> : k<0, 0>);
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF_BYTES);
    for (int rep = 0; rep < 2; ++rep)
      hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), 2 * BUF_BYTES, 0, wsl, xsl, out, st, nstage, x_wg_stride, sc, 5, c.mfma_mask, c.wslabs, c.xshared, c.posmode);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), 2 * BUF_BYTES, 0, wsl, xsl, out, st, nstage, x_wg_stride, sc, 5, c.mfma_mask, c.wslabs, c.xshared, c.posmode);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)nwg * 16);
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> per, blk;
    for (int i = 0; i < nwg * 8; ++i) {
      per.push_back((double)h[2 * i] / nstage);
      if (h[2 * i + 1]) blk.push_back((double)h[2 * i + 1] / nstage);
    }
    if (blk.empty()) blk.push_back(0);
    std::sort(per.begin(), per.end()); std::sort(blk.begin(), blk.end());
    const double gbps = 56.0 * 1024 * nstage * nwg / (ms * 1e-3) / 1e9;
    printf("%-66s | %7.3f ms | cycles/stage %6.0f (p90 %6.0f) | issuing waves blocked per stage %6.0f (p90 %6.0f) | LDS-DMA %6.0f GB/s chip = %5.1f B/clk/CU at 2.1 GHz\n",
           c.name, ms, per[per.size() / 2], per[per.size() * 9 / 10], blk[blk.size() / 2], blk[blk.size() * 9 / 10], gbps, gbps / 256 / 2.1);
    fflush(stdout);
  }
  return 0;
}
