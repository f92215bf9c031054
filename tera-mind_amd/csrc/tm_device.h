// Device-side helpers shared by the kernel translation units (not part of the public ABI).
#pragma once
#include "tm_kernels.h"
#include <math.h>
#include <atomic>

namespace tmk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define TM_EPS 1e-6f

// Host-side once-per-DEVICE flag for lazy launch setup (hipFuncSetAttribute applies to the current device's code
// object, so a per-process `static bool` would leave every device but the first without its LDS limit).  Safe to use from
// several host threads that drive different devices / streams: the bits are set with an atomic OR (a set-up that two
// threads both find undone is simply performed twice; hipFuncSetAttribute is idempotent).  A failing hipGetDevice, or a
// device index past the 128 the flag words cover, reports "not done" and marks nothing: the set-up then runs on every
// launch and ITS hip call returns the error.
struct DevOnce {
  std::atomic<unsigned long long> done[2] = {{0ull}, {0ull}};      // devices 0..127
  static int dev() { int d = -1; return hipGetDevice(&d) == hipSuccess ? d : -1; }
  bool need() const {
    const int d = dev();
    if (d < 0 || d >= 128) return true;
    return !((done[d >> 6].load(std::memory_order_acquire) >> (d & 63)) & 1ull);
  }
  void mark() {
    const int d = dev();
    if (d >= 0 && d < 128) done[d >> 6].fetch_or(1ull << (d & 63), std::memory_order_release);
  }
};
#ifndef TM_H16_T
#define TM_H16_T __bf16          // 16-bit float type of the y_h / gate_h tensors in this translation unit (tm_conv_bf16.hip)
#endif

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }
// tanh-GELU on the hardware exp2 / rcp: 0.5 x (1 + tanh u) = x / (1 + e^{-2u}),  u = sqrt(2/pi) (x + 0.044715 x^3);
// -2u log2(e) = x (k0 + k1 x^2): two multiplies, one FMA and one add around the two transcendentals (libm tanhf is ~50
// instructions; the epilogue of fc1 handles 4C values per voxel).  ~3e-7 relative on the result.
__device__ __forceinline__ float gelu_tanh_hw(float x) {
  constexpr float k0 = -2.0f * 0.7978845608028654f * 1.4426950408889634f;
  constexpr float k1 = k0 * 0.044715f;
  const float e = __builtin_amdgcn_exp2f(x * __builtin_fmaf(x * x, k1, k0));
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}
// SiLU for results that are rounded to a 16-bit type right away: v_exp_f32 and v_rcp_f32 (1 ulp each).  NOT `__frcp_rn`, which
// is an IEEE division on this compiler (v_div_scale x2, v_rcp, four FMAs, v_div_fmas, v_div_fixup: ten VALU instructions per
// value; the block-input passes and the fused conv epilogue were VALU-bound on it, profiles/r03_silu_rcp.txt)
__device__ __forceinline__ float silu_h16(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
__device__ __forceinline__ float gelu_tanh_f(float x) {
  const float kBeta = 0.7978845608028654f;   // sqrt(2/pi)
  const float kKappa = 0.044715f;
  float inner = kBeta * (x + kKappa * x * x * x);
  return 0.5f * x * (1.0f + tanhf(inner));
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ==========================================================================================
// Conv3d as implicit GEMM on the fp32 MFMA.
//   D[cout 32][voxel 32] += W[cout 32][k 2] * X[k 2][voxel 32]   (v_mfma_f32_32x32x2_f32)
// The weight is the A operand and the activation the B operand, so that each lane ends up
// with 4 consecutive couts of ONE voxel per accumulator quad: the CB8 store is a coalesced
// float4 per lane (32 voxels x 32 B contiguous per wave instruction).
// One ds_read_b128 of 4 channels feeds 4 MFMAs: lanes 0-31 carry channels {0..3}, lanes
// 32-63 channels {4..7} of the 8-channel block, identically for W and X, so MFMA #kk
// contracts channels {kk, 4+kk}.
//
// Replaces nn.Conv3d(k=3, padding=1) of ResBlock (reference model/MBAblocks.py:146-148,
// 182-186).  Z is 2 there, so for every output plane one of the three z taps only ever
// multiplies zero padding: a workgroup owns ONE output plane zo and runs 18 of the 27 taps
// (input planes zi = 0,1 with kz = zi + 1 - zo).
// ==========================================================================================
struct ConvArgs {
  const float* x; long x_nstride; long x_plane;
  const float* w; const float* bias;
  float* y; long y_nstride; long y_plane; int Cob;
  const float* res; long res_nstride;
  const float* gate; long gate_nstride;
  int N, S, Z, Cbi, ntile, flags;
  uint16_t* y_h = nullptr;          // optional bf16 CB8 output INSTEAD of y (strides in elements)
  long yh_nstride = 0;
  const uint16_t* gate_h = nullptr; // optional bf16 CB8 gate instead of `gate` (same plane geometry as y)
  long gate_h_nstride = 0;
  int Zin = 0, zoff = 0;            // conv3d NZI = 3: source plane = zo + zi + zoff, zero outside [0, Zin)
  const uint16_t* res_h = nullptr;  // 16-bit kernels only: 16-bit CB8 residual instead of `res` (the 16-bit activation stream)
  long res_h_nstride = 0;
  // gate tensor at HALF the in-plane resolution ([N][Cob][Z][S/2][S/2][8], read at (z, y >> 1, x >> 1)): the adaLN gates are
  // Linear(SiLU(cond)) of a nearest-x2 upsampled RNA level, i.e. constant over 2 x 2 voxel blocks.  gate_ls = log2(S), 0 = off
  int gate_ls = 0;
  int res_ls = 0;                   // the same for the fp32 residual `res` (3x3x3 conv epilogue): = log2(S), 0 = off
};

// in-plane element offset of voxel (z, y, x) -> (z, y >> 1, x >> 1) of the half-resolution plane; ls = log2(S)
__device__ __forceinline__ int half_res_off(int off8, int ls) {
  const int v = off8 >> 3, S1 = (1 << ls) - 1;
  const int x = v & S1, y = (v >> ls) & S1, z = v >> (2 * ls);
  return (((z << (ls - 1)) + (y >> 1)) << (ls - 1) | (x >> 1)) << 3;
}

// XCD-aware workgroup id (cdna guide T1): the dispatcher deals consecutive blockIdx round-robin over the 8 XCDs
// (private L2 each); this bijective remap hands every XCD one CONTIGUOUS run of logical tile ids, so the
// workgroups that share an activation tile (its other cout tiles), a halo or a weight slab hit the same L2.
__device__ __forceinline__ int xcd_swizzle(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// acc[ct][mt]: cout tile ct (32 couts) x voxel tile mt; cob0 = first 8-cout block of acc[0]
template <int WM, int WN = 2>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[WN][WM], int nt, int h,
                                              const int (&on)[WM], const int (&ooff)[WM], int S_out) {
  // ooff: in-plane float offset of the voxel in the OUTPUT plane geometry (or -1)
#pragma unroll
  for (int ct = 0; ct < WN; ++ct) {
    // (1) the residual / gate quads of this 32-cout slice of the wave tile are requested up front with UNCONDITIONAL loads
    //     (lanes without a valid voxel / cout block read element 0 of the tensor): 4 * WM (x2 with a gate) independent
    //     16-byte loads in flight instead of as many load -> use -> store chains behind per-voxel branches
    f32x4 rv[4][WM], gv[4][WM];
    bool ok[4][WM];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int cob = nt * (4 * WN) + ct * 4 + g;
#pragma unroll
      for (int mt = 0; mt < WM; ++mt) {
        ok[g][mt] = cob < a.Cob && ooff[mt] >= 0;
        const long pl = ok[g][mt] ? (long)cob * a.y_plane + ooff[mt] + 4 * h : 0;
        if (a.res) {
          const long rpl = !a.res_ls ? pl : (ok[g][mt] ? (long)cob * (a.y_plane >> 2) + half_res_off(ooff[mt], a.res_ls) + 4 * h : 0);
          rv[g][mt] = *(const f32x4*)(a.res + (ok[g][mt] ? (long)on[mt] * a.res_nstride : 0) + rpl);
        }
        if (a.gate) {
          const long gpl = !a.gate_ls ? pl : (ok[g][mt] ? (long)cob * (a.y_plane >> 2) + half_res_off(ooff[mt], a.gate_ls) + 4 * h : 0);
          gv[g][mt] = *(const f32x4*)(a.gate + (ok[g][mt] ? (long)on[mt] * a.gate_nstride : 0) + gpl);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int cob = nt * (4 * WN) + ct * 4 + g;
      if (cob >= a.Cob) continue;
      const f32x4 bv = *(const f32x4*)(a.bias + (long)cob * 8 + 4 * h);
#pragma unroll
      for (int mt = 0; mt < WM; ++mt) {
        if (!ok[g][mt]) continue;
        f32x4 o;
        o[0] = acc[ct][mt][4 * g + 0] + bv[0];
        o[1] = acc[ct][mt][4 * g + 1] + bv[1];
        o[2] = acc[ct][mt][4 * g + 2] + bv[2];
        o[3] = acc[ct][mt][4 * g + 3] + bv[3];
        if (a.flags & EPI_GELU) {
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = gelu_tanh_hw(o[j]);
        }
        const long pl = (long)cob * a.y_plane + ooff[mt] + 4 * h;
        if (a.gate) {
          o *= gv[g][mt];
        } else if (a.gate_h) {
          typedef TM_H16_T bf16x4_g __attribute__((ext_vector_type(4)));
          const bf16x4_g gb = *(const bf16x4_g*)(a.gate_h + (long)on[mt] * a.gate_h_nstride + pl);
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] *= (float)gb[j];
        }
        if (a.res) o = rv[g][mt] + o;
        if (a.y_h) {
          typedef TM_H16_T bf16x4_t __attribute__((ext_vector_type(4)));
          bf16x4_t ob;
#pragma unroll
          for (int j = 0; j < 4; ++j) ob[j] = (TM_H16_T)o[j];
          *(bf16x4_t*)(a.y_h + (long)on[mt] * a.yh_nstride + pl) = ob;
          continue;
        }
        float* yp = a.y + (long)on[mt] * a.y_nstride + pl;
        if (a.flags & EPI_UP2) {
          // nearest x2 on (H, W): ooff already addresses (2y, 2x) of the 2S plane
          *(f32x4*)(yp) = o;
          *(f32x4*)(yp + 8) = o;
          *(f32x4*)(yp + (long)S_out * 8) = o;
          *(f32x4*)(yp + (long)S_out * 8 + 8) = o;
        } else {
          *(f32x4*)(yp) = o;
        }
      }
    }
  }
}

}  // namespace tmk
