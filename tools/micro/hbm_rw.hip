// Microbenchmark: what does HBM give a streaming kernel as a function of its read : write mix?  The 16-bit glue of the tile
// step (block-input passes: 1 : 1; fc1 of the AttnBlock MLP: 1 : 4; fc2: 4 : 1) is priced against "6.3 TB/s achievable",
// a read figure.  Kernels: every lane moves 16-byte pieces, a workgroup owns contiguous 4 KB (256 lanes) chunks,
// grid-stride over a buffer far larger than L2 + MALL.
//   mode r : w  = R loads of 16 B per W stores of 16 B (R, W in {0..4}); loads are summed into the stored value
//   hipcc --offload-arch=gfx950 -O3 -o hbm_rw hbm_rw.hip && ./hbm_rw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int R, int W, bool NT>
__global__ __launch_bounds__(256) void k(const u32x4* __restrict__ src, u32x4* __restrict__ dst, long chunks) {
  // chunk = 256 lanes x 16 B; a workgroup handles chunk ids bid, bid + grid, ...; per chunk id it reads R chunks (from R
  // disjoint regions of src) and writes W chunks (to W disjoint regions of dst)
  const int lane = threadIdx.x;
  for (long c = blockIdx.x; c < chunks; c += gridDim.x) {
    u32x4 v[R > 0 ? R : 1];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const u32x4* p = src + ((long)r * chunks + c) * 256 + lane;
      v[r] = NT ? __builtin_nontemporal_load(p) : *p;
    }
    u32x4 s = {(unsigned)c, 1u, 2u, 3u};
#pragma unroll
    for (int r = 0; r < R; ++r) s += v[r];
    if (W == 0) {
      if (s[0] == 0x12345678u && s[1] == 0x9abcdef0u) dst[lane] = s;      // never true: keeps the loads
    }
#pragma unroll
    for (int w = 0; w < W; ++w) {
      u32x4* q = dst + ((long)w * chunks + c) * 256 + lane;
      u32x4 o = s; o[3] += (unsigned)w;
      if (NT) __builtin_nontemporal_store(o, q); else *q = o;
    }
  }
}

template <int R, int W, bool NT>
static void run(const u32x4* src, u32x4* dst, long chunks, int grid) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  std::vector<float> ms;
  for (int it = 0; it < 6; ++it) {
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((k<R, W, NT>), dim3(grid), dim3(256), 0, 0, src, dst, chunks);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float t; hipEventElapsedTime(&t, a, b);
    if (it) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  const double bytes = (double)(R + W) * chunks * 4096.0;
  printf("  r:w %d:%d %s grid %6d  %8.3f ms  %7.1f GB/s total  (read %7.1f, write %7.1f)\n", R, W, NT ? "nt" : "  ", grid, ms[ms.size() / 2],
         bytes / ms[ms.size() / 2] * 1e-6, R * chunks * 4096.0 / ms[ms.size() / 2] * 1e-6, W * chunks * 4096.0 / ms[ms.size() / 2] * 1e-6);
  hipEventDestroy(a); hipEventDestroy(b);
}

int main() {
  const long region = 1L << 30;                  // bytes per region; up to 4 regions per side
  const long chunks = region / 4096;
  u32x4 *src, *dst;
  if (hipMalloc(&src, 4 * region) != hipSuccess || hipMalloc(&dst, 4 * region) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(src, 1, 4 * region); hipMemset(dst, 0, 4 * region);
  hipDeviceSynchronize();
  for (int grid : {2048, 8192, 65536}) {
    printf("grid %d (256 threads per workgroup, 1 GiB per region)\n", grid);
    run<1, 0, false>(src, dst, chunks, grid);
    run<4, 0, false>(src, dst, chunks, grid);
    run<0, 1, false>(src, dst, chunks, grid);
    run<0, 4, false>(src, dst, chunks, grid);
    run<0, 4, true>(src, dst, chunks, grid);
    run<1, 1, false>(src, dst, chunks, grid);
    run<1, 1, true>(src, dst, chunks, grid);
    run<2, 2, false>(src, dst, chunks, grid);
    run<1, 4, false>(src, dst, chunks, grid);
    run<1, 4, true>(src, dst, chunks, grid);
    run<4, 1, false>(src, dst, chunks, grid);
    run<2, 1, false>(src, dst, chunks, grid);
    run<1, 2, false>(src, dst, chunks, grid);
  }
  hipFree(src); hipFree(dst);
  return 0;
}
