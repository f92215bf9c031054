// bf16 variant of the 3x3x3 implicit-GEMM conv (BASELINE config 4: bf16 weights / conv inputs,
// fp32 accumulate, fp32 residual stream).
//
//   D[cout 32][voxel 32] += W[cout 32][k 16] * X[k 16][voxel 32]      v_mfma_f32_32x32x16_bf16
//
// One MFMA contracts a PAIR of 8-channel blocks for one tap: lanes 0-31 carry block 2p, lanes
// 32-63 block 2p+1 (8 bf16 = one 16-byte ds_read_b128 per operand per lane).  The LDS images are
// byte-for-byte the shape of the fp32 kernel's ([tap][cout][32 B], [voxel][32 B]) but hold twice
// the channels, and an MFMA retires them 16x faster -- so the design point moves from "keep the
// matrix pipe fed with instructions" to "amortise the staged bytes":
//   * 8 waves, wave tile 64 cout x 128 voxels (2 x 4 MFMA tiles, 128 accumulator VGPRs),
//     workgroup tile 128 x 512 (Cout >= 128) or 64 x 1024 (Cout = 64): 12 B/clk/CU of staging
//     instead of 25 for the fp32 tile shape;
//   * a stage is one (channel-block pair, input plane): 9 taps, weights 9 x TN x 32 B, halo tile
//     of ONE plane; two LDS buffers, next stage's global loads are in flight during the MFMAs,
//     one barrier per stage;
//   * packed weights carry the 16-byte slot swap for couts with bit 3 set that makes the
//     ds_read_b128 of the weight fragment bank-conflict free (cdna guide T2 / microarch LDS table).
// Same operand orientation (weight = A, activation = B) and epilogue as conv3d_mfma.
//
// This translation unit is built TWICE (Makefile): as is for bf16, and with -DTM_H16_F16 for IEEE half (fp16) operands
// -- the arithmetic the reference itself runs on GPUs (autocast('cuda', fp16), diffusion/base.py:377; config_parm.py:40).
// The f16 MFMA forms take the same cycles as the bf16 ones; only the element type, the MFMA builtin, the host-side
// rounding of the packed weights and the exported names (…_f16 instead of …_bf16) differ.
#ifdef TM_H16_F16
#define TM_H16_T _Float16
#define TM_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define conv27_bf16 conv27_f16
#define conv27_pp conv27_pp_f16
#define conv1_bf16 conv1_f16
#define window_attn_bf16 window_attn_f16
#define window_attn_long window_attn_long_f16
#define fused_norm_epilogue fused_norm_epilogue_f16
#define launch_conv27_bf16 launch_conv27_f16
#define launch_conv1_bf16 launch_conv1_f16
#define launch_window_attn_bf16 launch_window_attn_f16
#define conv_bf16_pack_host conv_f16_pack_host
#define conv_bf16_pack_ups_host conv_f16_pack_ups_host
#define conv1_bf16_pack_host conv1_f16_pack_host
#else
#define TM_H16_T __bf16
#define TM_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#endif
#include "tm_device.h"

#include <stdlib.h>
#include <type_traits>
#include <string.h>

namespace tmk {

typedef TM_H16_T h16_t;
typedef h16_t bf16x8 __attribute__((ext_vector_type(8)));      // 8 x 16-bit floats (bf16 or, with TM_H16_F16, fp16)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // native 16-byte vector (HIP's uint4 class resists SROA)

// SiLU / tanh-GELU on the hardware exp2 / rcp (1 ulp each) for results that are rounded to a 16-bit type right away
__device__ __forceinline__ float silu_fast(float x) { return silu_h16(x); }
__device__ __forceinline__ float gelu_tanh_fast(float x) { return gelu_tanh_hw(x); }

struct ConvArgsH {
  ConvArgs c;                 // x / w are reinterpreted: x = bf16 CB8 (elements), w = bf16 packed
  long x_nstride_e, x_plane_e;   // in bf16 elements
  int Cbp;                    // channel-block pairs
  // fused ResBlock mid-section (only when the workgroup holds every cout of its voxels, n-tile count 1):
  // out_layers[0] RMSNorm(C) * w -> x(1+scale)+shift -> SiLU (MBAblocks.py:196-203,356-367) written as the bf16
  // input of the second conv; the fp32 conv output itself is not stored.
  int fuse;
  int bid0;                   // conv27: first tile id of this launch (a launch may cover a sub-range of the layer's tiles)
  const float* norm_w; const float* mod_scale; const float* mod_shift;
  long mod_stride; int per_image; float inv_c;
  uint16_t* a2; long a2_nstride;
};
// conv1 only: the input as a channel concat of up to three 16-bit CB8 tensors, read in place (the ResBlock skip conv:
// th.cat((h, skip, rna), 1) and to_collage are address rules of the x staging, never materialised).  Virtual channel
// block vb of the concat lives in source s = [vb >= cb1] + [vb >= cb2] at block vb - cb_s; a collaged source is read at
// the half-patch-shifted position of the (p1 x p2) source patch grid (model/unet_ours.py:325-341).
struct ConcatX {
  int nsrc;                       // 0: single tensor `c.x` (x_nstride_e)
  const uint16_t *p0, *p1p, *p2p; long ns0, ns1, ns2; int cb1, cb2; int col0, col1, col2;
  int cbtot, p1, p2;
};

template <int WNW>
__device__ __forceinline__ void fused_norm_epilogue(const ConvArgsH& ah, f32x16 (&acc)[2][4], int wn, int wm, int i32, int h,
                                                    const int (&on)[4], const int (&ooff)[4], float* red) {
  const ConvArgs& a = ah.c;
  float ss[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) ss[mt] = 0.f;
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 bv = *(const f32x4*)(a.bias + (long)(wn * 8 + ct * 4 + g) * 8 + 4 * h);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v = acc[ct][mt][4 * g + j] + bv[j];
          acc[ct][mt][4 * g + j] = v;
          ss[mt] += v * v;
        }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) ss[mt] += __shfl_xor(ss[mt], 32, 64);       // the voxel's other 4-cout halves
  if (WNW == 2) {                                                              // ... and its other 64 couts
    if (h == 0) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) red[((wm * 4 + mt) * 2 + wn) * 32 + i32] = ss[mt];
    }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) ss[mt] += red[((wm * 4 + mt) * 2 + (wn ^ 1)) * 32 + i32];
  }
  // apply pass with 16-byte stores: one v_permlane32_swap per register gives lane h = 0 the whole CB8 block 2q and lane
  // h = 1 the whole block 2q + 1 of its voxel (see conv_epilogue_h16 below)
  typedef h16_t h16x8_t __attribute__((ext_vector_type(8)));
  float rstd[4];
  long mo[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    rstd[mt] = 1.0f / sqrtf(ss[mt] * ah.inv_c + TM_EPS);
    mo[mt] = (long)(on[mt] / ah.per_image) * ah.mod_stride;
  }
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int cob = wn * 8 + ct * 4 + 2 * q + h;
      const f32x4 w0 = *(const f32x4*)(ah.norm_w + cob * 8), w1 = *(const f32x4*)(ah.norm_w + cob * 8 + 4);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // (by value first: __builtin_bit_cast applied directly to an ext-vector ELEMENT reads element 0 on this clang)
          const float xq = acc[ct][mt][8 * q + j], yq = acc[ct][mt][8 * q + 4 + j];
          const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(xq), __float_as_uint(yq), false, false);
          v[j] = __uint_as_float(r[0]);
          v[4 + j] = __uint_as_float(r[1]);
        }
        if (ooff[mt] < 0) continue;
        const float* scp = ah.mod_scale + mo[mt] + cob * 8;
        const float* shp = ah.mod_shift + mo[mt] + cob * 8;
        const f32x4 sc0 = *(const f32x4*)scp, sc1 = *(const f32x4*)(scp + 4), sh0 = *(const f32x4*)shp, sh1 = *(const f32x4*)(shp + 4);
        h16x8_t ob;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float a0 = w0[j] * (v[j] * rstd[mt]), a1 = w1[j] * (v[4 + j] * rstd[mt]);
          a0 = a0 * (1.0f + sc0[j]) + sh0[j];
          a1 = a1 * (1.0f + sc1[j]) + sh1[j];
          ob[j] = (h16_t)silu_fast(a0);
          ob[4 + j] = (h16_t)silu_fast(a1);
        }
        *(h16x8_t*)(ah.a2 + (long)on[mt] * ah.a2_nstride + (long)cob * a.y_plane + ooff[mt]) = ob;
      }
      __builtin_amdgcn_sched_barrier(0);          // keep the per-(cout block) loads from being hoisted en masse
    }
}

// Epilogue of the 16-bit kernels with 16-byte global accesses.  After the MFMAs lane (i32, h) holds, per accumulator quad
// g, couts 8g + 4h .. 8g + 4h + 3 of its voxel: HALF of a CB8 channel block, the other half sits in lane i32 + 32.  One
// v_permlane32_swap per register hands lane h = 0 the whole block 2q and lane h = 1 the whole block 2q + 1 (h = 0 gives its
// quad of block 2q + 1 away and receives the partner's quad of block 2q), so that a 16-bit CB8 entry (8 couts = 16 bytes)
// is read (gate, residual) and written by ONE lane: half the memory instructions of the 8-byte-per-lane form, each at the
// full 16-byte width.  Arithmetic and its order are those of conv_epilogue: (acc + bias) -> GELU -> * gate -> res + .
template <bool GATE>
__device__ __forceinline__ void conv_epilogue_h16(const ConvArgs& a, f32x16 (&acc)[2][4], int cob0, int h,
                                                  const int (&on)[4], const int (&ooff)[4]) {
  typedef h16_t h16x8 __attribute__((ext_vector_type(8)));
  // Without a gate (the 3x3x3 conv: the fragment registers of the K loop are dead here) ALL sixteen residual entries of the
  // wave tile are requested before the first one is used: the epilogue of the 64-cout level-0 layers moves 2 GB per launch
  // and is otherwise four load -> use -> store rounds, each exposing a full HBM round trip.
  h16x8 rall[GATE ? 1 : 4][4];
  if (!GATE && a.res_h) {
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int cob = cob0 + (g4 >> 1) * 4 + 2 * (g4 & 1) + h;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const bool okk = cob < a.Cob && ooff[mt] >= 0;
        const long rpl = !okk ? 0 : (!a.res_ls ? (long)cob * a.y_plane + ooff[mt]
                                               : (long)cob * (a.y_plane >> 2) + half_res_off(ooff[mt], a.res_ls));
        rall[GATE ? 0 : g4][mt] = *(const h16x8*)(a.res_h + (okk ? (long)on[mt] * a.res_h_nstride : 0) + rpl);
      }
    }
  }
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int cob = cob0 + ct * 4 + 2 * q + h;            // the block this lane owns after the swap
      // (1) the 16-bit gate / residual entries of this quarter of the wave tile are requested up front with UNCONDITIONAL
      //     loads (lanes without a valid voxel / cout block read element 0 of the tensor): four (eight with a gate)
      //     independent 16-byte loads in flight instead of load -> use -> store chains behind per-voxel branches
      h16x8 rb[4], gb[4];
      bool ok[4];
      long pl[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        ok[mt] = cob < a.Cob && ooff[mt] >= 0;
        pl[mt] = ok[mt] ? (long)cob * a.y_plane + ooff[mt] : 0;
        if (!GATE && a.res_h) rb[mt] = rall[GATE ? 0 : ct * 2 + q][mt];
        else if (a.res_h) {
          const long rpl = !a.res_ls ? pl[mt] : (ok[mt] ? (long)cob * (a.y_plane >> 2) + half_res_off(ooff[mt], a.res_ls) : 0);
          rb[mt] = *(const h16x8*)(a.res_h + (ok[mt] ? (long)on[mt] * a.res_h_nstride : 0) + rpl);
        }
        if (GATE && a.gate_h) {
          const long gpl = !a.gate_ls ? pl[mt] : (ok[mt] ? (long)cob * (a.y_plane >> 2) + half_res_off(ooff[mt], a.gate_ls) : 0);
          gb[mt] = *(const h16x8*)(a.gate_h + (ok[mt] ? (long)on[mt] * a.gate_h_nstride : 0) + gpl);
        }
      }
      f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
      if (cob < a.Cob) { b0 = *(const f32x4*)(a.bias + (long)cob * 8); b1 = *(const f32x4*)(a.bias + (long)cob * 8 + 4); }
      __builtin_amdgcn_sched_barrier(0);
      // (2) swap, bias, GELU, gate, residual, round, store
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // (by value first: __builtin_bit_cast applied directly to an ext-vector ELEMENT reads element 0 on this clang)
          const float xq = acc[ct][mt][8 * q + j], yq = acc[ct][mt][8 * q + 4 + j];
          const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(xq), __float_as_uint(yq), false, false);
          o[j] = __uint_as_float(r[0]) + b0[j];
          o[4 + j] = __uint_as_float(r[1]) + b1[j];
        }
        if (a.flags & EPI_GELU) {
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = a.y_h ? gelu_tanh_fast(o[j]) : gelu_tanh_f(o[j]);
        }
        if (GATE && a.gate_h) {
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] *= (float)gb[mt][j];
        } else if (GATE && a.gate) {
          if (ok[mt]) {
            const float* gp = a.gate + (long)on[mt] * a.gate_nstride + pl[mt];
            const f32x4 g0 = *(const f32x4*)gp, g1 = *(const f32x4*)(gp + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { o[j] *= g0[j]; o[4 + j] *= g1[j]; }
          }
        }
        if (a.res_h) {
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = (float)rb[mt][j] + o[j];
        } else if (a.res) {
          if (ok[mt]) {
            const float* rp = a.res + (long)on[mt] * a.res_nstride + pl[mt];
            const f32x4 r0 = *(const f32x4*)rp, r1 = *(const f32x4*)(rp + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { o[j] = r0[j] + o[j]; o[4 + j] = r1[j] + o[4 + j]; }
          }
        }
        if (!ok[mt]) continue;
        if (a.y_h) {
          h16x8 ob;
#pragma unroll
          for (int j = 0; j < 8; ++j) ob[j] = (h16_t)o[j];
          *(h16x8*)(a.y_h + (long)on[mt] * a.yh_nstride + pl[mt]) = ob;
        } else {
          float* yp = a.y + (long)on[mt] * a.y_nstride + pl[mt];
          *(f32x4*)yp = f32x4{o[0], o[1], o[2], o[3]};
          *(f32x4*)(yp + 4) = f32x4{o[4], o[5], o[6], o[7]};
        }
      }
    }
  }
}

// Epilogue of the Linears whose output is a 16-bit stream tensor (every conv1 launch of the 16-bit model): the same arithmetic
// as conv_epilogue_h16<true>, in the order  ALL loads -> arithmetic -> ALL stores.  vmcnt counts loads and stores in one
// in-order counter, so in the round-by-round form (load a quarter of the wave tile's gate / residual entries, use, store, next
// quarter) every round's loads waited for the previous round's stores to be ACKNOWLEDGED by memory -- four write round trips in
// series (stamps: 24 k cycles for the gated epilogue against 12 k for a plain one).  Here no store is issued before the last
// load has returned: the rounded 16-byte results take the place of the accumulators they came from, and the sixteen stores
// leave back to back at the end.  KIND: 1 = plain, 2 = tanh-GELU, 3 = 16-bit gate (full or half resolution) and 16-bit
// residual (the in-place update x <- x + gate * Linear(.): each entry is read and written by the same lane).
template <int KIND>
__device__ __forceinline__ void conv1_epilogue_stream(const ConvArgs& a, f32x16 (&acc)[2][4], int cob0, int h,
                                                      const int (&on)[4], const int (&ooff_in)[4]) {
  typedef h16_t h16x8 __attribute__((ext_vector_type(8)));
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  h16x8 rb[4], gb[4];
  // the voxel offsets are laundered through a VGPR here, after the K loop: otherwise every address of the epilogue (loop
  // invariant) is computed in front of the K loop and spilled across it
  int ooff[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) { ooff[mt] = ooff_in[mt]; asm volatile("" : "+v"(ooff[mt])); }
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    if (KIND == 3) {
      const int cob = cob0 + (rr >> 1) * 4 + 2 * (rr & 1) + h;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const bool ok = cob < a.Cob && ooff[mt] >= 0;
        const long pl = ok ? (long)cob * a.y_plane + ooff[mt] : 0;
        const long gpl = !a.gate_ls ? pl : (ok ? (long)cob * (a.y_plane >> 2) + half_res_off(ooff[mt], a.gate_ls) : 0);
        rb[mt] = *(const h16x8*)(a.res_h + (ok ? (long)on[mt] * a.res_h_nstride : 0) + pl);
        gb[mt] = *(const h16x8*)(a.gate_h + (ok ? (long)on[mt] * a.gate_h_nstride : 0) + gpl);
      }
    }
    {
      const int ct = rr >> 1, q = rr & 1;
      const int cob = cob0 + ct * 4 + 2 * q + h;            // the block this lane owns after the swap
      f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
      if (cob < a.Cob) { b0 = *(const f32x4*)(a.bias + (long)cob * 8); b1 = *(const f32x4*)(a.bias + (long)cob * 8 + 4); }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float xq = acc[ct][mt][8 * q + j], yq = acc[ct][mt][8 * q + 4 + j];
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(xq), __float_as_uint(yq), false, false);
          o[j] = __uint_as_float(sw[0]) + b0[j];
          o[4 + j] = __uint_as_float(sw[1]) + b1[j];
        }
        if (KIND == 2) {
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = gelu_tanh_fast(o[j]);
        }
        if (KIND == 3) {
#pragma unroll
          for (int j = 0; j < 8; ++j) { o[j] *= (float)gb[mt][j]; o[j] = (float)rb[mt][j] + o[j]; }
        }
        // the rounded 16-byte entry takes the place of the first four of the eight accumulator registers it came from
        h16x8 ob;
#pragma unroll
        for (int j = 0; j < 8; ++j) ob[j] = (h16_t)o[j];
        const f32x4v pk = __builtin_bit_cast(f32x4v, ob);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[ct][mt][8 * q + j] = pk[j];
      }
    }
    // one round's eight loads at a time: hoisted en masse they need 128 registers on top of the accumulators and spill (and a
    // second buffer set for round r + 1 does too: 256 VGPRs + scratch, measured 4 x slower than the round-by-round form)
    if (KIND == 3) __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int cob = cob0 + (rr >> 1) * 4 + 2 * (rr & 1) + h;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
      if (cob < a.Cob && ooff[mt] >= 0) {
        const int ct = rr >> 1, q = rr & 1;
        const f32x4v pk = {acc[ct][mt][8 * q], acc[ct][mt][8 * q + 1], acc[ct][mt][8 * q + 2], acc[ct][mt][8 * q + 3]};
        *(f32x4v*)(a.y_h + (long)on[mt] * a.yh_nstride + (long)cob * a.y_plane + ooff[mt]) = pk;
      }
  }
}

// Kind 3 of the stream epilogue (16-bit gate at full or half resolution + 16-bit residual, x <- x + gate * Linear(.)).  Order
//   L0 C0 L1 S0 C1 L2 S1 C2 L3 S2 C3 S3   (L = a round's eight loads, C = its arithmetic, S = its four stores):
// round r's stores leave BEHIND round r + 1's loads, so the wait in front of C(r+1) is vmcnt(4) -- it never waits for a store to
// be acknowledged (the round-by-round form L C S L C S ... did, four write round trips in series).  Holding one round's rounded
// results (16 registers) is all it costs; the forms that keep every result to the end or double-buffer the loads need more
// than 256 VGPRs next to the accumulators and spill (measured 4 x slower).  Addresses: per-voxel byte pointer, computed once
// per voxel tile, + cout block x plane bytes.
__device__ __forceinline__ void conv1_epilogue_gated(const ConvArgs& a, f32x16 (&acc)[2][4], int cob0, int h,
                                                     const int (&on)[4], const int (&ooff_in)[4]) {
  typedef h16_t h16x8 __attribute__((ext_vector_type(8)));
  const char* rbase[4];
  const char* gbase[4];
  char* ybase[4];
  bool okv[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    int oo = ooff_in[mt], nn = on[mt];
    asm volatile("" : "+v"(oo), "+v"(nn));            // after the K loop: keeps the address arithmetic from being hoisted over it
    okv[mt] = oo >= 0;
    const long n = okv[mt] ? nn : 0;
    const int o1 = okv[mt] ? oo : 0;
    rbase[mt] = (const char*)(a.res_h + n * a.res_h_nstride + o1);
    gbase[mt] = (const char*)(a.gate_h + n * a.gate_h_nstride + (a.gate_ls ? half_res_off(o1, a.gate_ls) : o1));
    ybase[mt] = (char*)(a.y_h + n * a.yh_nstride + o1);
  }
  const long ypl = (long)a.y_plane * 2, gpl = a.gate_ls ? ypl >> 2 : ypl;        // bytes per cout block
  h16x8 rb[4], gb[4], ob[4];
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    if (r < 4) {
      const int cob = min(cob0 + (r >> 1) * 4 + 2 * (r & 1) + h, a.Cob - 1);      // a block past the end re-reads the last one
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        rb[mt] = *(const h16x8*)(rbase[mt] + cob * ypl);
        gb[mt] = *(const h16x8*)(gbase[mt] + cob * gpl);
      }
    }
    if (r >= 1) {                                     // S(r - 1), behind L(r)
      const int cob = cob0 + ((r - 1) >> 1) * 4 + 2 * ((r - 1) & 1) + h;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        if (cob < a.Cob && okv[mt]) *(h16x8*)(ybase[mt] + cob * ypl) = ob[mt];
    }
    __builtin_amdgcn_sched_barrier(0);
    if (r < 4) {                                      // C(r)
      const int ct = r >> 1, q = r & 1;
      const int cob = min(cob0 + ct * 4 + 2 * q + h, a.Cob - 1);
      const f32x4 b0 = *(const f32x4*)(a.bias + (long)cob * 8), b1 = *(const f32x4*)(a.bias + (long)cob * 8 + 4);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float xq = acc[ct][mt][8 * q + j], yq = acc[ct][mt][8 * q + 4 + j];
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(xq), __float_as_uint(yq), false, false);
          o[j] = __uint_as_float(sw[0]) + b0[j];
          o[4 + j] = __uint_as_float(sw[1]) + b1[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { o[j] *= (float)gb[mt][j]; o[j] = (float)rb[mt][j] + o[j]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) ob[mt][j] = (h16_t)o[j];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Activation image in LDS: two arrays (k-half 0 / 1) of 16-byte slots [patch][halo row][pitch]; a lane reads
// slot(vox) of array h.  ds_read_b128 is served in the 16-lane groups G1 = {0-3,12-15,20-27} and
// G2 = {4-11,16-19,28-31} (MI355X_MICROARCH.md, LDS table) and is conflict-free when a group's 16 slots are
// distinct mod 16.  The MFMA column -> voxel map is free, so it is chosen per tile width:
//   TW = 32: column i -> (row, i): G1 and G2 each cover 16 distinct residues of one row (natural map);
//   TW = 16: G1 -> row 0, G2 -> row 1 of the 2-row MFMA tile;
//   TW =  8: pitch 12, G1 -> rows 0 and 2, G2 -> rows 1 and 3 of the 4-row MFMA tile.
// UPS: the 3x3x3 conv of a nearest-x2 UPSAMPLED input (ResBlock(up=True), model/MBAblocks.py:254-258) computed on the
// low-resolution tensor, as conv3d_mfma's UPS form (tm_kernels.hip): output phase (py, px) of low-resolution voxel (y, x)
// is a conv with 2 x 2 in-plane taps over (y + py - 1 .. y + py, x + px - 1 .. x + px) and pre-summed weights.  Four taps
// per plane would make a stage too short to cover its own LDS-DMA, so a stage is (channel-block pair, BOTH input planes):
// 8 taps, two halo images.
template <int TN, int TW, int NWV = 8, bool UPS = false>
struct HGeo {
  static constexpr int NT = NWV * 64;                 // threads: 8 waves (128 x 512 / 64 x 1024 tile) or 4 (half the voxels:
                                                      // twice the workgroups for launches that would leave CUs idle)
  static constexpr int WNW = TN / 64;                 // waves along cout
  static constexpr int WMW = NWV / WNW;               // waves along voxels
  static constexpr int TM = WMW * 128;                // voxels per workgroup
  static constexpr int TR = (TM / TW < TW) ? (TM / TW) : TW;
  static constexpr int NPB = TM / (TR * TW);
  static constexpr int HR = TR + 2, HC = TW + 2;
  static constexpr int HCP = (TW == 8) ? 12 : HC;     // slot pitch of a halo row
  static constexpr int XS = NPB * HR * HCP;           // slots of ONE k-half of ONE input plane
  static constexpr int XSP = (XS + 63) / 64 * 64;     // k-half arrays start on a wave's 64-slot boundary (LDS-DMA)
  static constexpr int NPL = UPS ? 2 : 1;             // input planes per stage
  static constexpr int NTAPS = UPS ? 8 : 9;           // taps per stage
  static constexpr int XPIECES = XSP * 2 * NPL;       // piece i -> LDS slot WPIECES + i  ([plane][k-half][slot])
  static constexpr int PX = (XPIECES + NT - 1) / NT;
  static constexpr int WPIECES = NTAPS * TN * 2;
  static constexpr int PW = (WPIECES + NT - 1) / NT;
  static constexpr int BUF16 = WPIECES + 2 * XSP * NPL;   // 16-byte units per LDS buffer
  static constexpr int LDS_BYTES = 2 * BUF16 * 16;
};

// MFMA column (0..31) of tile T of the workgroup -> (patch, row, col) inside the workgroup's region
template <int TW, int TR>
__device__ __forceinline__ void col_to_vox(int T, int i32, int& ps, int& r, int& c) {
  int g2, pos;
  if (i32 < 4) { g2 = 0; pos = i32; } else if (i32 < 12) { g2 = 1; pos = i32 - 4; }
  else if (i32 < 16) { g2 = 0; pos = i32 - 8; } else if (i32 < 20) { g2 = 1; pos = i32 - 8; }
  else if (i32 < 28) { g2 = 0; pos = i32 - 12; } else { g2 = 1; pos = i32 - 16; }
  if (TW == 32) { ps = T / TR; r = T % TR; c = i32; }
  else if (TW == 16) { ps = T / (TR / 2); r = (T % (TR / 2)) * 2 + g2; c = pos; }
  else { ps = T / (TR / 4); r = (T % (TR / 4)) * 4 + g2 + 2 * (pos >> 3); c = pos & 7; }
}

#define TM_GLDS16(gptr, lptr)                                                                  \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),      \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

// the same piece as `buffer_load_dwordx4 v_off, s[rsrc], s_off offen lds`: M0 = LDS base, per-lane byte offset + scalar byte offset
#define TM_BLDS16(rsrc, voff, soff, lptr)                                                      \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lptr), 16, (int)(voff), (int)(soff), 0, 0)

#ifndef TM_ABL
#define TM_ABL 0      // diagnostic builds: 64 = two 16x16x32 MFMAs per 32x32x16 (timing only), 1 = no stage barrier, 8 = barrier without waiting for the LDS-DMAs, 2 = no fragment ds_reads after a stage's first tap, 4 = no LDS-DMA after stage 0
#endif
#if TM_ABL & 64
// timing experiment (results wrong): each 32x32x16 MFMA of the conv27 main loops issued as two 16x16x32 MFMAs on the same
// operand registers (equal FLOPs, 4-pass instructions) -- what the in-loop clock does with the smaller instruction
typedef float f32x4_e __attribute__((ext_vector_type(4)));
#ifdef TM_H16_F16
#define TM_MFMA16X(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#else
#define TM_MFMA16X(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#endif
__device__ __forceinline__ f32x16 mfma_main(bf16x8 a, bf16x8 b, f32x16 c) {
  f32x4_e c0 = {c[0], c[1], c[2], c[3]}, c1 = {c[4], c[5], c[6], c[7]};
  c0 = TM_MFMA16X(a, b, c0);
  c1 = TM_MFMA16X(a, b, c1);
  c[0] = c0[0]; c[1] = c0[1]; c[2] = c0[2]; c[3] = c0[3];
  c[4] = c1[0]; c[5] = c1[1]; c[6] = c1[2]; c[7] = c1[3];
  return c;
}
#define TM_MFMA16_MAIN(a, b, c) mfma_main(a, b, c)
#else
#define TM_MFMA16_MAIN(a, b, c) TM_MFMA16(a, b, c)
#endif
#ifdef TM_STAMPS
// Diagnostic build only (make diag -> libteramind_hip_diag.so, tools/conv27_stamps.py): wave 0 of every workgroup records
// s_memtime at kernel entry, main-loop entry, main-loop exit and kernel exit into a buffer of its own.
__device__ unsigned long long* g_tm_stamps = nullptr;      // [capacity][16]: t0 t1 t2 t3 grid bid tag realtime + 8 kernel-specific words
__device__ unsigned int g_tm_stamp_cap = 0;
__device__ unsigned int g_tm_stamp_next = 0;
#define TM_STAMP(i) do { if (stamp_slot) stamp_slot[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TM_STAMP(i) do { } while (0)
#endif

template <int TN, int TW, bool FUSE, int NWV = 8, bool UPS = false>
__global__ __launch_bounds__(NWV * 64, 2) void conv27_bf16(ConvArgsH ah) {
  using G = HGeo<TN, TW, NWV, UPS>;
  constexpr int NTAPS = G::NTAPS;
  constexpr int NT = G::NT;
  const ConvArgs& a = ah.c;
  extern __shared__ __attribute__((aligned(16))) u32x4 lds16[];
#ifdef TM_STAMPS
  unsigned long long* stamp_slot = nullptr;
  if (threadIdx.x == 0 && g_tm_stamps) {
    const unsigned int k = atomicAdd(&g_tm_stamp_next, 1u);
    if (k < g_tm_stamp_cap) {
      stamp_slot = g_tm_stamps + (size_t)k * 16;
      stamp_slot[4] = gridDim.x; stamp_slot[5] = blockIdx.x;
      stamp_slot[6] = (unsigned long long)ah.Cbp * 1000000ull + (unsigned long long)(TN * 1000 + TW * 10 + (FUSE ? 1 : 0) + (UPS ? 2 : 0));
      stamp_slot[7] = __builtin_amdgcn_s_memrealtime();
    }
  }
  TM_STAMP(0);
#endif

  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int wvu = __builtin_amdgcn_readfirstlane(wv);
  const int i32 = lane & 31, h = lane >> 5;
  const int wn = wv % G::WNW, wm = wv / G::WNW;

  const int S = a.S;
  const int tiles_c = S / TW, tiles_r = S / G::TR;
  const int tiles = tiles_c * tiles_r;
  const int bid = ah.bid0 + xcd_swizzle(blockIdx.x, gridDim.x);
  const int nt = bid % a.ntile;                        // n-tile of TN couts
  int mt_ = bid / a.ntile;
  int py = 0, px = 0;                                  // UPS: output phase (the four phases of a tile are grid neighbours)
  if (UPS) { py = (mt_ >> 1) & 1; px = mt_ & 1; mt_ >>= 2; }
  const int pg = mt_ / (a.Z * tiles);                  // a.Z output (= input) planes, pad 1 in z
  mt_ -= pg * a.Z * tiles;
  const int zo = mt_ / tiles;
  mt_ -= zo * tiles;
  const int tr = mt_ / tiles_c, tc = mt_ - tr * tiles_c;

  const h16_t* xg = (const h16_t*)a.x;
  const h16_t* wg = (const h16_t*)a.w;

  // ---- staging descriptors: piece i = tid + k*NT lands in LDS slot WPIECES + i (lane-linear per wave,
  //      as the LDS-DMA requires); halo slots outside the plane / patch are never loaded and keep the
  //      zeros written once below ----
  // Pieces are issued as `buffer_load_dwordx4 ... offen lds`: a per-lane 32-bit byte offset that never changes (xvo[], wvo)
  // plus a scalar offset per stage, against two descriptors built once -- the packed weights of this n-tile (phase), and the
  // activations of this workgroup's patch group.  A piece then costs two scalar adds (M0, the scalar offset) and the load
  // itself instead of ~25 instructions of 64-bit address arithmetic per piece (200 per wave per stage, issued between the
  // MFMAs).  Lanes whose halo slot lies outside the plane / patch carry an offset past the descriptor's range: the
  // hardware drops them (the slot keeps / gets the zeros of the padding).
  unsigned xvo[G::PX];
#pragma unroll
  for (int k = 0; k < G::PX; ++k) {
    const int i = tid + k * NT;
    unsigned off = 0x80000000u;                       // past any descriptor range, and no 32-bit wrap when the scalar offset is added
    if (i < G::XPIECES) {
      const int pl = i / (2 * G::XSP);                 // plane of the stage (UPS: both input planes; otherwise 0)
      const int half = (i - pl * 2 * G::XSP) / G::XSP;
      int v = i - (pl * 2 + half) * G::XSP;
      if (v < G::XS) {
        const int hc = v % G::HCP; v /= G::HCP;
        const int hr = v % G::HR;
        const int ps = v / G::HR;
        const int n = pg * G::NPB + ps;
        const int y = tr * G::TR + hr - 1, x = tc * TW + hc - 1;
        if (hc < G::HC && n < a.N && y >= 0 && y < S && x >= 0 && x < S)
          off = (unsigned)(((long)ps * ah.x_nstride_e + (long)half * ah.x_plane_e + (long)pl * S * S * 8 + ((long)y * S + x) * 8) * 2);
      }
    }
    xvo[k] = off;
  }
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(xg + (long)pg * G::NPB * ah.x_nstride_e), 0, (int)((long)G::NPB * ah.x_nstride_e * 2), 0x00020000);
  // packed weights: [n-tile][pair][kz 3][9 taps][TN][2][8]; UPS: [phase][n-tile][pair][kz 3][4 taps][TN][2][8]
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(UPS ? wg + ((long)(py * 2 + px) * a.ntile + nt) * ah.Cbp * 12 * TN * 16 : wg + (long)nt * ah.Cbp * 27 * TN * 16), 0,
      ah.Cbp * (UPS ? 12 : 27) * TN * 32, 0x00020000);
  const int wvo = tid * 16;

  // ---- fragment addresses (16-byte units inside a buffer) ----
  int xb[4], on[4], ooff[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    int ps, r, c;
    col_to_vox<TW, G::TR>(wm * 4 + mt, i32, ps, r, c);
    xb[mt] = G::WPIECES + h * G::XSP + (ps * G::HR + r) * G::HCP + c;
    const int n = pg * G::NPB + ps;
    on[mt] = n;
    const int y = tr * G::TR + r, x = tc * TW + c;
    if (UPS) ooff[mt] = (n < a.N) ? ((zo * 2 * S + 2 * y + py) * 2 * S + 2 * x + px) * 8 : -1;
    else ooff[mt] = (n < a.N) ? ((zo * S + y) * S + x) * 8 : -1;
  }
  const int wb = (wn * 64 + i32) * 2 + (h ^ ((i32 >> 3) & 1));
  // fragment offset of tap t inside a buffer: weights t * TN * 2; activations relative to xb[]
  auto tap_xd = [&](int t) __attribute__((always_inline)) {
    return UPS ? (t >> 2) * 2 * G::XSP + (((t >> 1) & 1) + py) * G::HCP + (t & 1) + px : (t / 3) * G::HCP + (t % 3);
  };

  f32x16 acc[2][4];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ct][mt][r] = 0.f;

  // (no zero fill of the activation images: out-of-range lanes of `buffer_load ... lds` write zeros, see conv27_pp)

  // one stage = (channel-block pair, input plane zi): global -> LDS by LDS-DMA, no staging registers.  Output plane zo
  // reads the input planes [zo-1, zo+1] that exist (kz = zi + 1 - zo): 2 of 3 for the z_size-2 checkpoint model, 1 for
  // z_size 1, 2 or 3 for z_size 4 / 8 -- planes in the zero padding are never staged nor multiplied
  const int zi0 = UPS ? 0 : (zo > 0 ? zo - 1 : 0);
  const int npl = UPS ? 1 : (zo + 2 < a.Z ? zo + 2 : a.Z) - zi0;      // stages per channel-block pair
  // LDS-DMA piece p (0 .. PW + PX - 1) of stage hs: the weight pieces first, then the halo-tile pieces
  auto issue_piece = [&](int hs, int p) __attribute__((always_inline)) {
    const int cbp = hs / npl, zi = zi0 + hs % npl;
    u32x4* base = lds16 + (hs & 1) * G::BUF16;
    if (p < G::PW) {
      const int k = p;
      // UPS (Z == 2): the stage's planes 0, 1 meet kz = 1 - zo, 2 - zo: eight consecutive taps of the phase's 12
      const int ws = UPS ? (cbp * 3 + (1 - zo)) * 4 * TN * 32 : (cbp * 3 + (zi + 1 - zo)) * 9 * TN * 32;
      if (G::WPIECES % NT == 0 || k * NT + wvu * 64 < G::WPIECES) TM_BLDS16(wrs, wvo, ws + k * NT * 16, base + k * NT + wvu * 64);
    } else {
      const int k = p - G::PW;
      const int xs = (cbp * 2 * (int)ah.x_plane_e + (UPS ? 0 : zi * S * S * 8)) * 2;
      if (G::XPIECES % NT == 0 || k * NT + wvu * 64 < G::XPIECES) TM_BLDS16(xrs, xvo[k], xs, base + G::WPIECES + k * NT + wvu * 64);
    }
  };
  constexpr int NP = G::PW + G::PX;                    // DMA instructions per wave per stage
  // The next stage's DMAs are not issued in one burst behind the barrier (every wave of the workgroup would then spend the
  // same ~100 cycles per instruction issuing them while all four matrix pipes idle) but PPT at a time behind the MFMA
  // groups of taps 0 .. 3: a wave's VMEM issue overlaps its own and its SIMD partner's MFMAs, and the last piece still
  // has five taps (~2.5k cycles) to land before the barrier.
  constexpr int PPT = (NP + 3) / 4;
  constexpr int PPH = (NP + 1) / 2;                    // pieces per tap when a wave issues its share behind two taps

  const int NH = npl * ah.Cbp;
#pragma unroll
  for (int p = 0; p < NP; ++p) issue_piece(0, p);
  __syncthreads();
  TM_STAMP(1);
  // Fragment registers: THREE sets.  Tap 0 of a stage uses set 2, taps t >= 1 use set (t - 1) & 1; tap t + 1's ds_reads
  // are issued before tap t's MFMAs.  The stage barrier sits in FRONT of tap 8's MFMAs, not behind them: once every wave
  // has its tap-8 fragments in registers the buffer is dead, so the waves synchronise there, request the next stage's
  // tap-0 fragments (set 2) from the other buffer right away and only then issue tap 8's eight MFMAs -- the LDS round
  // trip of a new stage's first reads (all eight waves hit the LDS at once) is covered by matrix work instead of
  // leaving the four matrix pipes idle behind every barrier.
  bf16x8 wf[3][2], xf[3][4];
  {
    const u32x4* buf0 = lds16;
    wf[2][0] = __builtin_bit_cast(bf16x8, buf0[wb]);
    wf[2][1] = __builtin_bit_cast(bf16x8, buf0[64 + wb]);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) xf[2][mt] = __builtin_bit_cast(bf16x8, buf0[xb[mt] + tap_xd(0)]);
  }
  for (int hs = 0; hs < NH; ++hs) {
    const bool more = hs + 1 < NH;
    const u32x4* buf = lds16 + (hs & 1) * G::BUF16;
    const u32x4* nbuf = lds16 + ((hs + 1) & 1) * G::BUF16;
#pragma unroll
    for (int tap = 0; tap < NTAPS; ++tap) {
      const int cur = tap == 0 ? 2 : ((tap - 1) & 1);
      const int nxt = tap & 1;                             // set of tap + 1 (taps 1 .. 8)
      // This tap's fragments were requested one tap ago, behind eight MFMAs: they have long landed.  Saying so HERE, before
      // the next tap's reads are issued, keeps hipcc from waiting lgkmcnt(0) after them (it does not emit a counted
      // lgkmcnt(6) in this loop), which would expose a full LDS round trip in front of every second MFMA group.
      __builtin_amdgcn_s_waitcnt(0xC07F);                  // lgkmcnt(0) only (vmcnt / expcnt fields = no wait)
      __builtin_amdgcn_sched_barrier(0);
      if (tap < NTAPS - 1) {
        const int t1 = tap + 1;
        const int xd = tap_xd(t1);
        if (!(TM_ABL & 2)) {
          wf[nxt][0] = __builtin_bit_cast(bf16x8, buf[t1 * TN * 2 + wb]);
          wf[nxt][1] = __builtin_bit_cast(bf16x8, buf[t1 * TN * 2 + 64 + wb]);
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) xf[nxt][mt] = __builtin_bit_cast(bf16x8, buf[xb[mt] + xd]);
        } else {
          wf[nxt][0] = wf[2][0]; wf[nxt][1] = wf[2][1];
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) xf[nxt][mt] = xf[2][mt];
        }
      } else {
        // every fragment of this buffer is in registers (the wait above); this wave's share of the next stage has landed
        // once its LDS-DMAs have drained (the vmcnt(0) of __syncthreads); behind the barrier everyone's has
        if (TM_ABL & 8) __builtin_amdgcn_s_barrier();          // diagnostic: barrier WITHOUT the vmcnt(0) drain of the LDS-DMAs
        else if (!(TM_ABL & 1)) __syncthreads();
        if (more) {
          wf[2][0] = __builtin_bit_cast(bf16x8, nbuf[wb]);
          wf[2][1] = __builtin_bit_cast(bf16x8, nbuf[64 + wb]);
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) xf[2][mt] = __builtin_bit_cast(bf16x8, nbuf[xb[mt] + tap_xd(0)]);
        }
      }
      // keep tap t+1's ds_reads above tap t's MFMAs (hipcc otherwise sinks them to just before their use and
      // every 8-MFMA group eats a full LDS round trip)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          acc[ct][mt] = TM_MFMA16_MAIN(wf[cur][ct], xf[cur][mt], acc[ct][mt]);
      __builtin_amdgcn_sched_barrier(0);
      if (more && tap < 4 && !(TM_ABL & 4)) {
        if (NWV == 8) {
          // Waves w and w + 4 share a SIMD and leave every barrier in lockstep.  An LDS-DMA instruction blocks its wave's
          // issue for 100-200 cycles; issued by both partners behind the same MFMA groups, those blocks coincide and the
          // matrix pipe idles for all of them (tools/conv27_stamps.py ablation: 6250 cycles per stage with, 4850 without
          // the DMAs, 4608 ideal).  So the halves take turns: waves 0-3 issue their pieces behind taps 0 and 1, waves 4-7
          // behind taps 2 and 3 -- a wave's DMA issue then runs under its partner's MFMAs.
          if ((tap >> 1) == (wv >> 2)) {
#pragma unroll
            for (int p = (tap & 1) * PPH; p < ((tap & 1) + 1) * PPH && p < NP; ++p) issue_piece(hs + 1, p);
          }
        } else {
#pragma unroll
          for (int p = tap * PPT; p < (tap + 1) * PPT && p < NP; ++p) issue_piece(hs + 1, p);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  TM_STAMP(2);
  if (FUSE) fused_norm_epilogue<G::WNW>(ah, acc, wn, wm, i32, h, on, ooff, (float*)lds16);
  else conv_epilogue_h16<false>(a, acc, (nt * G::WNW + wn) * 8, h, on, ooff);
#ifdef TM_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TM_STAMP(3);
#endif
}
// ---- the 8-wave form as a PING-PONG of its two wave groups ------------------------------------------------------------
// Same tiles, LDS images, operand maps and epilogues as conv27_bf16<.., 8, ..>; what changes is WHEN a wave does what.
// Waves 0-3 (group 0) and 4-7 (group 1) sit pairwise on the four SIMDs.  In the loop above both partners of a SIMD ran the
// same stream in lockstep: their ds_reads, their LDS-DMA issue (100-200 cycles of blocked issue each) and their MFMAs
// coincided, and the stage barrier made every wave pay for the slowest (75 % of the MFMA-ideal cycles; 96 % without the
// barrier, 95 % without the DMAs: profiles/r02_conv27_mainloop_ablation.txt).  Here the K loop is cut into SEGMENTS of
// SEGT taps (3 of a stage's 9; 2 of 8 in the upsampled-input form) and the groups alternate strictly, one barrier apart
// (the stagger barrier group 1 executes once): while one group issues the 8 * SEGT MFMAs of a segment back to back at
// raised priority, the other requests the fragments of ITS next segment (ds_read_b128 into the one fragment set a wave now
// needs) and issues its share of the next stage's LDS-DMAs; at the next barrier the roles swap.  A SIMD's matrix pipe is
// handed from one partner to the other with nothing but MFMAs in the handed-over stream, and every memory instruction of
// a wave is issued under its partner's MFMAs (cdna guide: the 8-phase template's two wave groups).
//
// Interval k (between workgroup barriers k and k+1): k = 2s: group 0 computes segment s, group 1 loads segment s;
// k = 2s + 1: group 0 loads segment s + 1, group 1 computes segment s.  LDS protocol (two stage buffers as before):
//   * a wave waits for its own ds_reads (lgkmcnt(0)) BEFORE the barrier that ends its load interval, so stage L - 1's
//     buffer is dead once both groups have passed the barrier behind their load of (L - 1, last segment), which is before
//     group 0's load of (L, segment 0): every load step of stage L may refill that buffer with pieces of stage L + 1;
//   * LDS-DMA is a throughput resource of its own -- a CU takes in 20-30 B/clk of it, and the 56 KB a stage needs are half
//     of a stage's MFMA time at that rate -- so the pieces are dealt EVENLY over the load steps (PPSched below), a few per
//     step, issued between the fragment reads of the step's taps (a wave that issues pieces back to back sits blocked at
//     the second for as long as the first takes, ~200 cycles by the stamps, with its ds_reads queued behind);
//   * ROLLING DRAIN: a wave begins every load step with vmcnt(0) -- that retires the pieces of its previous load step, two
//     intervals old and long landed -- so a piece issued in interval k is complete and, behind the barrier that ends
//     interval k + 2, visible to everyone from interval k + 3 on.  A piece first read in segment a of stage L + 1 (group
//     0's load of it is interval 2 ((L + 1) NSEG + a) - 1) may therefore be issued in the load step of (L, j) -- group 1's
//     is interval 2 (L NSEG + j) -- for j <= NSEG - 2 + a: the halo tile and the first segment's weights go into the first
//     NSEG - 1 steps, later segments' weights anywhere.
// Both groups execute the same number of barriers (group 1: the stagger barrier instead of the one behind its last MFMAs).
template <int TN, int TW, bool UPS>
struct PPSched {
  using G = HGeo<TN, TW, 8, UPS>;
  static constexpr int SEGT = UPS ? 2 : 3;             // taps per segment
  static constexpr int NSEG = G::NTAPS / SEGT;         // segments (= load steps) per stage
  static constexpr int NP = G::PW + G::PX;             // LDS-DMA pieces (wave-instructions) per wave per stage
  // first segment that reads piece p: weight piece k holds slots [k NT, (k + 1) NT) of [tap][TN][2]; a halo piece is read
  // from the first tap on (UPS: the pieces wholly inside plane 1 from tap 4 on)
  __host__ __device__ static constexpr int first_seg(int p) {
    if (p < G::PW) return (p * G::NT) / (TN * 2) / SEGT;
    return (UPS && (p - G::PW) * G::NT >= 2 * G::XSP) ? 4 / SEGT : 0;
  }
  __host__ __device__ static constexpr int deadline(int p) {
    const int d = NSEG - 2 + first_seg(p);
    return d < NSEG - 1 ? d : NSEG - 1;
  }
  // Greedy deal: pieces in order of deadline (halo pieces before weight pieces, then by index), each to the least loaded
  // step it may go to (ties: the earliest).
  __host__ __device__ static constexpr int step_of(int p) {
    int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int dl = 0; dl < NSEG; ++dl)
      for (int pass = 0; pass < 2; ++pass)
        for (int q = 0; q < NP; ++q) {
          if (deadline(q) != dl || ((q >= G::PW) != (pass == 0))) continue;
          int best = 0;
          for (int st = 1; st <= dl; ++st)
            if (cnt[st] < cnt[best]) best = st;
          if (q == p) return best;
          ++cnt[best];
        }
    return 0;
  }
  // piece[step][n]: the n-th piece (in index order) of a load step, cnt[step] of them
  struct Tbl { int piece[4][16]; int cnt[4]; };
  __host__ __device__ static constexpr Tbl make() {
    Tbl t = {};
    for (int st = 0; st < 4; ++st) {
      t.cnt[st] = 0;
      for (int n = 0; n < 16; ++n) t.piece[st][n] = -1;
    }
    for (int q = 0; q < NP; ++q) {
      const int st = step_of(q);
      t.piece[st][t.cnt[st]++] = q;
    }
    return t;
  }
};
template <int TN, int TW, bool FUSE, bool UPS = false>
__global__ __launch_bounds__(512, 2) void conv27_pp(ConvArgsH ah) {
  using G = HGeo<TN, TW, 8, UPS>;
  constexpr int NTAPS = G::NTAPS;
  constexpr int NT = G::NT;
  constexpr int SEGT = UPS ? 2 : 3;                    // taps per segment
  constexpr int NSEG = NTAPS / SEGT;                   // segments per stage
  static_assert(NSEG * SEGT == NTAPS && NSEG >= 2, "segment shape");
  const ConvArgs& a = ah.c;
  extern __shared__ __attribute__((aligned(16))) u32x4 lds16[];
#ifdef TM_STAMPS
  // diagnostic build: lane 0 of wave 0 (group 0) and of wave 4 (group 1) each own a 16-word slot: [0..3] entry / loop entry /
  // loop exit / exit, [4] grid [5] block [6] tag (+4: group 1) [7] realtime, [8..12] cycles summed over the segments:
  // MFMA issue | wait at the barrier behind it | LDS-DMA issue | ds_reads until landed | wait at the barrier before the MFMAs
  unsigned long long* stamp_slot = nullptr;
  if ((threadIdx.x & 255) == 0 && g_tm_stamps) {
    const unsigned int k = atomicAdd(&g_tm_stamp_next, 1u);
    if (k < g_tm_stamp_cap) {
      stamp_slot = g_tm_stamps + (size_t)k * 16;
      stamp_slot[4] = gridDim.x; stamp_slot[5] = blockIdx.x;
      stamp_slot[6] = (unsigned long long)ah.Cbp * 1000000ull + (unsigned long long)(TN * 1000 + TW * 10 + (FUSE ? 1 : 0) + (UPS ? 2 : 0) + (threadIdx.x ? 4 : 0));
      stamp_slot[7] = __builtin_amdgcn_s_memrealtime();
    }
  }
  unsigned long long pp_acc[5] = {0ull, 0ull, 0ull, 0ull, 0ull}, pp_prev = 0ull;
#define PP_STAMP(i_)                                                                                       \
  do {                                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    unsigned long long t_;                                                                                 \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                             \
    pp_acc[i_] += t_ - pp_prev; pp_prev = t_;                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
  } while (0)
  TM_STAMP(0);
#else
#define PP_STAMP(i_) do { } while (0)
#endif

  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int wvu = __builtin_amdgcn_readfirstlane(wv);  // provably wave-uniform: LDS-DMA bases and role branches stay scalar
  const int grp = wvu >> 2;
  const int i32 = lane & 31, h = lane >> 5;
  const int wn = wv % G::WNW, wm = wv / G::WNW;

  const int S = a.S;
  const int tiles_c = S / TW, tiles_r = S / G::TR;
  const int tiles = tiles_c * tiles_r;
  const int bid = ah.bid0 + xcd_swizzle(blockIdx.x, gridDim.x);
  const int nt = bid % a.ntile;
  int mt_ = bid / a.ntile;
  int py = 0, px = 0;
  if (UPS) { py = (mt_ >> 1) & 1; px = mt_ & 1; mt_ >>= 2; }
  const int pg = mt_ / (a.Z * tiles);
  mt_ -= pg * a.Z * tiles;
  const int zo = mt_ / tiles;
  mt_ -= zo * tiles;
  const int tr = mt_ / tiles_c, tc = mt_ - tr * tiles_c;

  const h16_t* xg = (const h16_t*)a.x;
  const h16_t* wg = (const h16_t*)a.w;

  // staging descriptors, fragment addresses, output coordinates: exactly those of conv27_bf16
  // staging descriptors as in conv27_bf16: buffer_load ... lds pieces, per-lane offsets fixed, scalar offset per stage
  unsigned xvo[G::PX];
#pragma unroll
  for (int k = 0; k < G::PX; ++k) {
    const int i = tid + k * NT;
    unsigned off = 0x80000000u;                       // past any descriptor range, and no 32-bit wrap when the scalar offset is added
    if (i < G::XPIECES) {
      const int pl = i / (2 * G::XSP);                 // plane of the stage (UPS: both input planes; otherwise 0)
      const int half = (i - pl * 2 * G::XSP) / G::XSP;
      int v = i - (pl * 2 + half) * G::XSP;
      if (v < G::XS) {
        const int hc = v % G::HCP; v /= G::HCP;
        const int hr = v % G::HR;
        const int ps = v / G::HR;
        const int n = pg * G::NPB + ps;
        const int y = tr * G::TR + hr - 1, x = tc * TW + hc - 1;
        if (hc < G::HC && n < a.N && y >= 0 && y < S && x >= 0 && x < S)
          off = (unsigned)(((long)ps * ah.x_nstride_e + (long)half * ah.x_plane_e + (long)pl * S * S * 8 + ((long)y * S + x) * 8) * 2);
      }
    }
    xvo[k] = off;
  }
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(xg + (long)pg * G::NPB * ah.x_nstride_e), 0, (int)((long)G::NPB * ah.x_nstride_e * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(UPS ? wg + ((long)(py * 2 + px) * a.ntile + nt) * ah.Cbp * 12 * TN * 16 : wg + (long)nt * ah.Cbp * 27 * TN * 16), 0,
      ah.Cbp * (UPS ? 12 : 27) * TN * 32, 0x00020000);
  const int wvo = tid * 16;
  int xb[4], on[4], ooff[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    int ps, r, c;
    col_to_vox<TW, G::TR>(wm * 4 + mt, i32, ps, r, c);
    xb[mt] = G::WPIECES + h * G::XSP + (ps * G::HR + r) * G::HCP + c;
    const int n = pg * G::NPB + ps;
    on[mt] = n;
    const int y = tr * G::TR + r, x = tc * TW + c;
    if (UPS) ooff[mt] = (n < a.N) ? ((zo * 2 * S + 2 * y + py) * 2 * S + 2 * x + px) * 8 : -1;
    else ooff[mt] = (n < a.N) ? ((zo * S + y) * S + x) * 8 : -1;
  }
  const int wb = (wn * 64 + i32) * 2 + (h ^ ((i32 >> 3) & 1));
  auto tap_xd = [&](int t) __attribute__((always_inline)) {
    return UPS ? (t >> 2) * 2 * G::XSP + (((t >> 1) & 1) + py) * G::HCP + (t & 1) + px : (t / 3) * G::HCP + (t % 3);
  };

  f32x16 acc[2][4];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ct][mt][r] = 0.f;

  // No zero fill of the activation images: a lane whose halo slot lies outside the plane / patch carries an out-of-range offset,
  // and an out-of-range lane of `buffer_load ... lds` WRITES ZEROS to its LDS slot (checked on gfx950 with the images prefilled
  // with NaNs: every bit-exact op test still passes) -- every slot a tap reads is rewritten by every stage.
  const int zi0 = UPS ? 0 : (zo > 0 ? zo - 1 : 0);
  const int npl = UPS ? 1 : (zo + 2 < a.Z ? zo + 2 : a.Z) - zi0;      // stages per channel-block pair
  const int NH = npl * ah.Cbp;
  // LDS-DMA piece p of the stage (cbp, zi) into the buffer at `base`: the weight pieces first, then the halo-tile pieces
  auto issue_piece = [&](u32x4* base, int cbp, int zi, int p) __attribute__((always_inline)) {
    if (p < G::PW) {
      const int k = p;
      const int ws = UPS ? (cbp * 3 + (1 - zo)) * 4 * TN * 32 : (cbp * 3 + (zi + 1 - zo)) * 9 * TN * 32;
      if (G::WPIECES % NT == 0 || k * NT + wvu * 64 < G::WPIECES) TM_BLDS16(wrs, wvo, ws + k * NT * 16, base + k * NT + wvu * 64);
    } else {
      const int k = p - G::PW;
      const int xs = (cbp * 2 * (int)ah.x_plane_e + (UPS ? 0 : zi * S * S * 8)) * 2;
      if (G::XPIECES % NT == 0 || k * NT + wvu * 64 < G::XPIECES) TM_BLDS16(xrs, xvo[k], xs, base + G::WPIECES + k * NT + wvu * 64);
    }
  };
  using SCH = PPSched<TN, TW, UPS>;
  constexpr int NP = SCH::NP;                          // DMA instructions per wave per stage
  constexpr typename SCH::Tbl sch = SCH::make();       // which pieces go with which load step
  static_assert(NP <= 16 && NSEG <= 4, "schedule table shape");

  // prologue: stage 0 into buffer 0, landed before the first read (stage 1 arrives through the load steps of stage 0)
#pragma unroll
  for (int p = 0; p < NP; ++p) issue_piece(lds16, 0, zi0, p);
  __syncthreads();                                     // vmcnt(0) + barrier
  TM_STAMP(1);
#ifdef TM_STAMPS
  pp_prev = __builtin_amdgcn_s_memtime();
#endif

  bf16x8 wf[SEGT][2], xf[SEGT][4];
  // coordinates of the stage whose pieces are being issued: the stage after the one whose segments are being loaded
  int f_cbp = (NH > 1) ? 1 / npl : 0, f_zi = zi0 + ((NH > 1) ? 1 % npl : 0);
  // load step of segment j: (dma) retire this wave's older pieces, then the segment's ds_reads from `rb` tap by tap with this
  // step's pieces of stage (f_cbp, f_zi) between them, into `db`
  auto load_step = [&](const u32x4* rb, int j, bool dma, u32x4* db) __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int t = 0; t < SEGT; ++t) {
      if (t < sch.cnt[j]) {
        if (dma) issue_piece(db, f_cbp, f_zi, sch.piece[j][t]);
      }
      const int tap = j * SEGT + t;
      const int xd = tap_xd(tap);
      wf[t][0] = __builtin_bit_cast(bf16x8, rb[tap * TN * 2 + wb]);
      wf[t][1] = __builtin_bit_cast(bf16x8, rb[tap * TN * 2 + 64 + wb]);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) xf[t][mt] = __builtin_bit_cast(bf16x8, rb[xb[mt] + xd]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (dma) {
#pragma unroll
      for (int n = SEGT; n < 16; ++n)
        if (n < sch.cnt[j]) issue_piece(db, f_cbp, f_zi, sch.piece[j][n]);
    }
  };
  load_step(lds16, 0, NH > 1, lds16 + G::BUF16);       // segment (0, 0) + the first pieces of stage 1
  __builtin_amdgcn_s_waitcnt(0xC07F);                  // lgkmcnt(0) only
      asm volatile("" ::: "memory");
  if (grp == 1) __builtin_amdgcn_s_barrier();          // the stagger: group 1 runs one interval behind group 0
  __builtin_amdgcn_sched_barrier(0);

  for (int hs = 0; hs < NH; ++hs) {
    const bool more = hs + 1 < NH;
    const bool fetch_cur = more;                       // load steps 1 .. of stage hs carry pieces of stage hs + 1
    const bool fetch_next = hs + 2 < NH;               // load step 0 of stage hs + 1 (end of this iteration): pieces of stage hs + 2
    u32x4* buf = lds16 + (hs & 1) * G::BUF16;
    u32x4* nbuf = lds16 + ((hs + 1) & 1) * G::BUF16;
#pragma unroll
    for (int j = 0; j < NSEG; ++j) {
      // ---- compute interval of segment (hs, j): fragments are in registers (waited for before the barrier) ----
      __builtin_amdgcn_s_waitcnt(0xC07F);                  // lgkmcnt(0) only
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      PP_STAMP(4);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int t = 0; t < SEGT; ++t)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
            acc[ct][mt] = TM_MFMA16_MAIN(wf[t][ct], xf[t][mt], acc[ct][mt]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      PP_STAMP(0);
      if (!(j == NSEG - 1 && !more && grp == 1)) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      PP_STAMP(1);
      // ---- load interval: the next segment's fragments + this step's pieces of the stage after the one being loaded ----
      if (j < NSEG - 1) {
        load_step(buf, j + 1, fetch_cur, nbuf);
      } else if (more) {
        if (++f_zi == zi0 + npl) { f_zi = zi0; ++f_cbp; }       // (f_cbp, f_zi) = stage hs + 2
        load_step(nbuf, 0, fetch_next, buf);
      }
#ifdef TM_STAMPS
      __builtin_amdgcn_s_waitcnt(0xC07F);
      PP_STAMP(3);
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  TM_STAMP(2);
#ifdef TM_STAMPS
  if (stamp_slot) { for (int i = 0; i < 5; ++i) stamp_slot[8 + i] = pp_acc[i]; }
#endif
  if (FUSE) fused_norm_epilogue<G::WNW>(ah, acc, wn, wm, i32, h, on, ooff, (float*)lds16);
  else conv_epilogue_h16<false>(a, acc, (nt * G::WNW + wn) * 8, h, on, ooff);
#ifdef TM_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TM_STAMP(3);
#endif
#undef PP_STAMP
}

// ==================================================================================================================
// conv27_pp16: the ping-pong kernel above on v_mfma_f32_16x16x32 (Z == 2, not the upsampled-input form).
//
// Why: with random operands the 4-pass 16x16x32 instruction sustains a higher shader clock than the 8-pass 32x32x16
// (profiles/r02_micro_mfma16_sustained.txt: 2.05-2.11 vs 1.82-1.87 GHz bare; issuing this kernel's MFMAs as 16x16x32 pairs
// on the same registers, timing only: +8.6 % over the tile-step layer list, profiles/r03_mfma16x16x32_in_loop.txt).
//
// K = 32 per instruction = the 16 channels of a stage x TWO taps.  A stage (channel-block pair, input plane) has 9 taps, so
// the loop runs over channel-block pairs with BOTH input planes of the pair: 18 taps = 9 UNITS of two taps.  Buffer X holds
// the pair's first plane, buffer Y its second; units 0-3 read taps (0,1) .. (6,7) of X, unit 4 STRADDLES (tap 8 of X with
// tap 0 of Y), units 5-8 read taps (1,2) .. (7,8) of Y.  One unit = 4 weight fragments (16 couts x 32 k) + 8 activation
// fragments (32 k x 16 voxels) = 12 ds_read_b128 and 32 MFMAs (512 matrix-pipe cycles) per wave; the two wave groups alternate
// unit by unit exactly as in conv27_pp (same barriers, same rolling vmcnt(0) drain).
// Lane (c16 = lane % 16, q = lane / 16) of a fragment holds k-group q: tap (q / 2) of the unit's two, channels 8 (q % 2) ..+8.
//   weights:  LDS slot tap * TN * 2 + cout * 2 + (q % 2), UNSWIZZLED (conflict-free for this read pattern: a 16-lane service
//             group reads couts {0-3, 12-15} of k-half 0 and {4-11} of k-half 1 -> 16 distinct slots mod 16); the packed global
//             weights keep conv27_bf16's swizzle (the 4-wave tail launches read the same arena), so the LDS-DMA lane that fills
//             slot (cout, k) fetches global slot (cout, k ^ (cout >> 3 & 1)): a per-lane constant offset, no extra work.
//   halo:     array (q % 2) at slot(voxel) + shift(tap); the MFMA column -> voxel map (col_to_vox16) puts the 16 voxels of a
//             tile on 16 distinct slots mod 16 for every tile width (TW = 8: 4 rows x 4 columns at pitch 12).
// DMA schedule per period of 9 load steps (PPSched16): a piece may be issued from the step after the last unit that reads the
// slots' old content, and at the latest two steps before the first unit that reads the new one; pieces for Y carry the
// CURRENT pair's second plane (window [0, first - 2]), pieces for X the NEXT pair's first plane ([last + 1, min(8, 7 + first)]).
// Epilogue: the 16x16 accumulators (lane: voxel c16, couts 4 q + j of tile a) are converted to the 32x32x16 register layout
// with one v_permlane16_swap + one v_permlane32_swap per register pair (lane bit 4 <-> tile-column bit, then lane bit 5 <->
// cout bit 3), after which conv_epilogue_h16 / fused_norm_epilogue run unchanged.
// ==================================================================================================================
#ifdef TM_H16_F16
#define TM_MFMA16K32_ACC(c, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
#define conv27_pp16 conv27_pp16_f16
#else
#define TM_MFMA16K32_ACC(c, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
#endif

template <int TW, int TR>
__device__ __forceinline__ void col_to_vox16(int T, int i32, int& ps, int& r, int& c) {
  if (TW == 32) { ps = T / TR; r = T % TR; c = i32; }
  else if (TW == 16) { ps = T / (TR / 2); r = (T % (TR / 2)) * 2 + (i32 >> 4); c = i32 & 15; }
  else { ps = T / (TR / 4); r = (T % (TR / 4)) * 4 + ((i32 & 15) >> 2); c = 4 * (i32 >> 4) + (i32 & 3); }
}

template <int TN, int TW>
struct PPSched16 {
  using G = HGeo<TN, TW, 8, false>;
  static constexpr int NP = G::PW + G::PX;
  static constexpr int NU = 9, MAXC = 6;
  __host__ __device__ static constexpr int tap_lo(int p) { return (p * G::NT) / (TN * 2); }
  __host__ __device__ static constexpr int tap_hi(int p) { const int t = ((p + 1) * G::NT - 1) / (TN * 2); return t > 8 ? 8 : t; }
  __host__ __device__ static constexpr int unit_x(int t) { return t / 2; }               // tap 8 -> the straddling unit 4
  __host__ __device__ static constexpr int unit_y(int t) { return 4 + (t + 1) / 2; }     // tap 0 -> unit 4
  // window of load steps in which piece p for target tgt (0: Y of this pair, 1: X of the next pair) may be issued
  __host__ __device__ static constexpr int lo(int tgt, int p) { return tgt == 0 ? 0 : (p < G::PW ? unit_x(tap_hi(p)) : 4) + 1; }
  __host__ __device__ static constexpr int hi(int tgt, int p) {
    if (tgt == 0) return (p < G::PW ? unit_y(tap_lo(p)) : 4) - 2;
    const int h = NU + (p < G::PW ? unit_x(tap_lo(p)) : 0) - 2;
    return h > NU - 1 ? NU - 1 : h;
  }
  struct Tbl { int piece[NU][MAXC]; int tgt[NU][MAXC]; int cnt[NU]; bool ok; };
  __host__ __device__ static constexpr Tbl make() {
    Tbl t = {};
    t.ok = true;
    for (int m = 0; m < NU; ++m) {
      t.cnt[m] = 0;
      for (int n = 0; n < MAXC; ++n) { t.piece[m][n] = -1; t.tgt[m][n] = 0; }
    }
    // by deadline, halo pieces first; each to the least loaded step of its window (ties: the earliest)
    for (int dl = 0; dl < NU; ++dl)
      for (int pass = 0; pass < 2; ++pass)
        for (int tg = 0; tg < 2; ++tg)
          for (int q = 0; q < NP; ++q) {
            if (hi(tg, q) != dl || ((q >= G::PW) != (pass == 0))) continue;
            if (lo(tg, q) > dl) { t.ok = false; continue; }
            int best = lo(tg, q);
            for (int st = lo(tg, q) + 1; st <= dl; ++st)
              if (t.cnt[st] < t.cnt[best]) best = st;
            if (t.cnt[best] >= MAXC) { t.ok = false; continue; }
            t.piece[best][t.cnt[best]] = q;
            t.tgt[best][t.cnt[best]] = tg;
            ++t.cnt[best];
          }
    return t;
  }
};

template <int TN, int TW, bool FUSE>
__global__ __launch_bounds__(512, 2) void conv27_pp16(ConvArgsH ah) {
  using G = HGeo<TN, TW, 8, false>;
  constexpr int NT = G::NT;
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  const ConvArgs& a = ah.c;
  extern __shared__ __attribute__((aligned(16))) u32x4 lds16[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int wvu = __builtin_amdgcn_readfirstlane(wv);
  const int grp = wvu >> 2;
  const int i32 = lane & 31, h = lane >> 5;                 // the epilogue's (32x32 layout) lane coordinates
  const int c16 = lane & 15, kh = (lane >> 4) & 1, hi2 = lane >> 5;   // fragment coordinates: column / row, k-half, which tap of the unit
  const int wn = wv % G::WNW, wm = wv / G::WNW;

  const int S = a.S;
  const int tiles_c = S / TW, tiles_r = S / G::TR;
  const int tiles = tiles_c * tiles_r;
  const int bid = ah.bid0 + xcd_swizzle(blockIdx.x, gridDim.x);
  const int nt = bid % a.ntile;
  int mt_ = bid / a.ntile;
  const int pg = mt_ / (a.Z * tiles);
  mt_ -= pg * a.Z * tiles;
  const int zo = mt_ / tiles;
  mt_ -= zo * tiles;
  const int tr = mt_ / tiles_c, tc = mt_ - tr * tiles_c;

  const h16_t* xg = (const h16_t*)a.x;
  const h16_t* wg = (const h16_t*)a.w;
  unsigned xvo[G::PX];
#pragma unroll
  for (int k = 0; k < G::PX; ++k) {
    const int i = tid + k * NT;
    unsigned off = 0x80000000u;
    if (i < G::XPIECES) {
      const int half = i / G::XSP;
      int v = i - half * G::XSP;
      if (v < G::XS) {
        const int hc = v % G::HCP; v /= G::HCP;
        const int hr = v % G::HR;
        const int ps = v / G::HR;
        const int n = pg * G::NPB + ps;
        const int y = tr * G::TR + hr - 1, x = tc * TW + hc - 1;
        if (hc < G::HC && n < a.N && y >= 0 && y < S && x >= 0 && x < S)
          off = (unsigned)(((long)ps * ah.x_nstride_e + (long)half * ah.x_plane_e + ((long)y * S + x) * 8) * 2);
      }
    }
    xvo[k] = off;
  }
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(xg + (long)pg * G::NPB * ah.x_nstride_e), 0, (int)((long)G::NPB * ah.x_nstride_e * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)(wg + (long)nt * ah.Cbp * 27 * TN * 16), 0,
                                                                       ah.Cbp * 27 * TN * 32, 0x00020000);
  const int wvo = (tid ^ ((tid >> 4) & 1)) * 16;            // de-swizzling fetch: LDS slot (cout, k) <- packed slot (cout, k ^ (cout >> 3 & 1))
  // fragment addresses = per-lane base + compile-time offset (the ds_read immediate).  A lane's tap is the unit's first (lanes
  // 0-31) or second (32-63), so the base carries hi2 * (offset of the second tap - offset of the first): three variants for
  // the halo (the taps of a unit are neighbours in a row: +1; the pair wraps to the next halo row: + HCP - 2; the straddling
  // unit: tap 8 of X -> tap 0 of Y), and for the weights (next tap: + TN * 2 slots, from buffer X or buffer Y; the straddle)
  // LDS regions (16-byte slots): [X weights | X halo | Y halo | Y weights] -- both halo images within one 16-bit ds_read offset
  // of a per-lane base
  constexpr int XH = G::WPIECES, YH = G::WPIECES + 2 * G::XSP, YW = G::WPIECES + 4 * G::XSP;
  static_assert(YW + G::WPIECES == 2 * G::BUF16, "LDS regions");
  constexpr int XD_ROW = 1, XD_WRAP = G::HCP - 2, XD_STRADDLE = (YH - XH) - (2 * G::HCP + 2);
  // (the wrap / straddle variants are the row variant + hi2 * const, added where they are used: three of the nine units).
  // The wave's eight 16-voxel tiles lie at compile-time slot distances from its first (col_to_vox16 is affine over a wave's
  // four 32-voxel tiles for every geometry), so ONE per-lane base register serves all eight fragment reads of a unit.
  int xb0;
  {
    int ps, r, c;
    col_to_vox16<TW, G::TR>(wm * 4, c16, ps, r, c);
    xb0 = G::WPIECES + kh * G::XSP + (ps * G::HR + r) * G::HCP + c + hi2 * XD_ROW;
  }
  auto tile_delta = [](int b) constexpr {                    // slot distance of tile b = 2 mt + bq from tile 0
    const int mt = b >> 1, bq = b & 1;
    if (TW == 32) return mt * G::HCP + 16 * bq;
    if (TW == 16) return (2 * mt + bq) * G::HCP;
    return ((mt >> 1) * G::HR + (mt & 1) * 4) * G::HCP + 4 * bq;
  };
  static_assert(TW != 32 || G::TR % 4 == 0, "tile map"); static_assert(TW != 16 || (G::TR / 2) % 4 == 0, "tile map");
  static_assert(TW != 8 || G::TR / 4 == 2, "tile map");
  const int wb = (wn * 64 + c16) * 2 + kh;
  const int wbA = wb + hi2 * TN * 2;                     // (the Y and straddle variants: wbA + a per-unit add, below)

  f32x4_t acc[4][8];
#pragma unroll
  for (int ca = 0; ca < 4; ++ca)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[ca][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // Z == 2: both input planes feed both output planes -- tap plane kz = zi + 1 - zo
  auto issue_piece = [&](int tgt, int cbp, int p) __attribute__((always_inline)) {
    u32x4* wbase = lds16 + (tgt == 0 ? YW : 0);
    u32x4* hbase = lds16 + (tgt == 0 ? YH : XH);
    const int zi = tgt == 0 ? 1 : 0;
    if (p < G::PW) {
      const int ws = (cbp * 3 + (zi + 1 - zo)) * 9 * TN * 32;
      if (G::WPIECES % NT == 0 || p * NT + wvu * 64 < G::WPIECES) TM_BLDS16(wrs, wvo, ws + p * NT * 16, wbase + p * NT + wvu * 64);
    } else {
      const int k = p - G::PW;
      const int xs = (cbp * 2 * (int)ah.x_plane_e + zi * S * S * 8) * 2;
      if (G::XPIECES % NT == 0 || k * NT + wvu * 64 < G::XPIECES) TM_BLDS16(xrs, xvo[k], xs, hbase + k * NT + wvu * 64);
    }
  };
  using SCH = PPSched16<TN, TW>;
  constexpr int NP = SCH::NP;
  constexpr typename SCH::Tbl sch = SCH::make();
  static_assert(sch.ok, "PPSched16: a piece has no admissible load step");

  // prologue: the first pair's plane 0 into buffer X (its plane 1 arrives through the load steps 0 .. 6 like every Y)
#pragma unroll
  for (int p = 0; p < NP; ++p) issue_piece(1, 0, p);
  __syncthreads();

  bf16x8 wf[4], xf[8];
  const int NPAIR = ah.Cbp;
  // load step of unit m: retire this wave's older pieces, then the unit's 12 fragment reads with this step's pieces between them;
  // cy / cx: the pair whose Y / X pieces are issued (cx < 0: none)
  auto load_step = [&](auto mc, int cy, int cx) __attribute__((always_inline)) {
    constexpr int m = decltype(mc)::value;
    constexpr int t1 = m < 4 ? 2 * m : (m == 4 ? 8 : 2 * (m - 4) - 1), t2 = m < 4 ? 2 * m + 1 : (m == 4 ? 0 : 2 * (m - 4));
    constexpr int b1 = m <= 4 ? 0 : YH - XH;
    constexpr int sh1 = (t1 / 3) * G::HCP + (t1 % 3), sh2 = (t2 / 3) * G::HCP + (t2 % 3);
    constexpr int xk = m == 4 ? 2 : (sh2 - sh1 == XD_ROW ? 0 : 1);           // which halo base variant
    static_assert(m == 4 || sh2 - sh1 == XD_ROW || sh2 - sh1 == XD_WRAP, "tap pair geometry");
    static_assert(m == 4 || t2 == t1 + 1, "tap pair");
    constexpr int xo = b1 + sh1;                                               // immediate part, in 16-byte slots
    constexpr int wo = t1 * TN * 2;                                            // (the Y weights' base is in wbY: the offset field is 16 bits)
    static_assert((xo + tile_delta(7) + 1) * 16 < 65536 && (wo + 3 * 32 + 1) * 16 < 65536, "ds_read offset field");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int xbu = xb0;
    if (xk != 0) {
      int d = hi2 * ((xk == 1 ? XD_WRAP : XD_STRADDLE) - XD_ROW);
      asm volatile("" : "+v"(d));                      // keeps the add inside the unit
      xbu += d;
    }
    int wd = m == 4 ? hi2 * (YW - 9 * TN * 2) : (m < 4 ? 0 : YW);
    if (m >= 4) asm volatile("" : "+v"(wd));
    const int wsel = wbA + wd + wo;
    auto pieces = [&](int n) __attribute__((always_inline)) {
      if (n < sch.cnt[m]) {
        const int tg = sch.tgt[m][n];
        if (tg == 0) issue_piece(0, cy, sch.piece[m][n]);
        else if (cx >= 0) issue_piece(1, cx, sch.piece[m][n]);
      }
    };
    pieces(0);
#pragma unroll
    for (int ca = 0; ca < 4; ++ca) wf[ca] = __builtin_bit_cast(bf16x8, lds16[wsel + ca * 32]);
    __builtin_amdgcn_sched_barrier(0);
    pieces(1);
#pragma unroll
    for (int b = 0; b < 4; ++b) xf[b] = __builtin_bit_cast(bf16x8, lds16[xbu + (xo + tile_delta(b))]);
    __builtin_amdgcn_sched_barrier(0);
    pieces(2);
#pragma unroll
    for (int b = 4; b < 8; ++b) xf[b] = __builtin_bit_cast(bf16x8, lds16[xbu + (xo + tile_delta(b))]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n = 3; n < SCH::MAXC; ++n) pieces(n);
  };
  load_step(std::integral_constant<int, 0>{}, 0, NPAIR > 1 ? 1 : -1);
  __builtin_amdgcn_s_waitcnt(0xC07F);                  // lgkmcnt(0) only
  asm volatile("" ::: "memory");
  if (grp == 1) __builtin_amdgcn_s_barrier();          // the stagger: group 1 runs one interval behind group 0
  __builtin_amdgcn_sched_barrier(0);

  for (int cb = 0; cb < NPAIR; ++cb) {
    const bool more = cb + 1 < NPAIR;
    const int cx = more ? cb + 1 : -1;
    auto unit = [&](auto mc) __attribute__((always_inline)) {
      constexpr int m = decltype(mc)::value;
      __builtin_amdgcn_s_waitcnt(0xC07F);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ca = 0; ca < 4; ++ca)
#pragma unroll
        for (int b = 0; b < 8; ++b) TM_MFMA16K32_ACC(acc[ca][b], wf[ca], xf[b]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      if (!(m == 8 && !more && grp == 1)) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (m < 8) load_step(std::integral_constant<int, (m < 8 ? m + 1 : 0)>{}, cb, cx);
      else if (more) load_step(std::integral_constant<int, 0>{}, cb + 1, cb + 2 < NPAIR ? cb + 2 : -1);
      __builtin_amdgcn_sched_barrier(0);
    };
    unit(std::integral_constant<int, 0>{}); unit(std::integral_constant<int, 1>{}); unit(std::integral_constant<int, 2>{});
    unit(std::integral_constant<int, 3>{}); unit(std::integral_constant<int, 4>{}); unit(std::integral_constant<int, 5>{});
    unit(std::integral_constant<int, 6>{}); unit(std::integral_constant<int, 7>{}); unit(std::integral_constant<int, 8>{});
  }

  // The MFMAs above are inline asm with the accumulator TIED (dst = src C): left to the register allocator, the 32 four-register
  // accumulators wandered through the whole file and the LDS-DMA offsets were spilled to scratch inside the loop (each reload
  // an s_waitcnt vmcnt(0) that also waits for the pieces just issued).  The hazard recogniser does not see inside inline asm:
  // the last MFMA's result needs its passes before a VALU instruction may read it.
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
  // output coordinates of the epilogue's lanes (computed here: eight registers the loop does not have to carry)
  int on[4], ooff[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    int ps, r, c;
    col_to_vox16<TW, G::TR>(wm * 4 + mt, i32, ps, r, c);
    const int n = pg * G::NPB + ps;
    on[mt] = n;
    const int y = tr * G::TR + r, x = tc * TW + c;
    ooff[mt] = (n < a.N) ? ((zo * S + y) * S + x) * 8 : -1;
  }
  // 16x16 tiles -> the 32x32x16 accumulator layout of the epilogues
  f32x16 acc32[2][4];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int ap = 0; ap < 2; ++ap)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float xv = acc[2 * ct + ap][2 * mt][j], yv = acc[2 * ct + ap][2 * mt + 1][j];
          const auto s1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(xv), __float_as_uint(yv), false, false);
          const auto s2 = __builtin_amdgcn_permlane32_swap(s1[0], s1[1], false, false);
          acc32[ct][mt][8 * ap + j] = __uint_as_float(s2[0]);
          acc32[ct][mt][8 * ap + 4 + j] = __uint_as_float(s2[1]);
        }
  if (FUSE) fused_norm_epilogue<G::WNW>(ah, acc32, wn, wm, i32, h, on, ooff, (float*)lds16);
  else conv_epilogue_h16<false>(a, acc32, (nt * G::WNW + wn) * 8, h, on, ooff);
}

#ifdef TM_STAMPS
#ifndef TM_H16_F16
extern "C" int tm_diag_stamps(unsigned long long* dev_buf, unsigned int capacity) {     // dev_buf == null: disable
  unsigned int zero = 0;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_tm_stamps), &dev_buf, sizeof(dev_buf)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_tm_stamp_cap), &capacity, sizeof(capacity)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_tm_stamp_next), &zero, sizeof(zero)) != hipSuccess) return -1;
  return 0;
}
extern "C" int tm_diag_stamp_count(void) {
  unsigned int n = 0;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_tm_stamp_next), sizeof(n)) != hipSuccess) return -1;
  return (int)n;
}
#endif
#endif

// ---- 1x1x1 conv / Linear on '(z h w) c' tokens, bf16 operands (flat voxel tiles) -----------
// No tap reuse: at the matrix-pipe rate the staging traffic is 4096*(1/TN + 1/TM) = 40 B/clk/CU, i.e. about
// 100 KB must be in flight per CU to cover an L2/HBM round trip.  Structure: a ring of NB LDS buffers filled
// by LDS-DMA, NB-1 stages in flight, ONE raw s_barrier per stage and a COUNTED s_waitcnt vmcnt so that the
// younger stages stay in flight across the barrier (cdna guide, "Pipelining across barriers").  Every
// LDS-DMA instruction is issued unconditionally -- pieces that do not exist (voxels past the end, channel
// pairs past Cbp, ring slots past the last stage) read a zero page -- so each wave issues exactly LPS
// instructions per stage and the vmcnt immediate is a compile-time constant.
// NWV = waves per workgroup: 8 (one 128 x 512 / 64 x 1024 workgroup per CU) or 4 (two co-resident 128 x 256 / 64 x 512
// workgroups per CU: the K loops here are 4..64 stages long, and with a single workgroup per CU nothing covers its ring
// fill, drain and epilogue; with two, one's tail overlaps the other's MFMAs).
// a / b for 0 <= a < 2^21, 0 < b, rb = 1.0f / b: float multiply + one correction step instead of the ~40-instruction integer
// division (quarter-rate multiplies); `small` false (wave-uniform): the plain division
__device__ __forceinline__ int idiv_small(int a, int b, float rb, bool small) {
  if (!small) return a / b;
  int q = (int)((float)a * rb);
  const int r = a - q * b;
  q += (r >= b ? 1 : 0) - (r < 0 ? 1 : 0);
  return q;
}

template <int TN, int NWV>
struct H1Geo {
  static constexpr int NT = NWV * 64;
  static constexpr int WNW = TN / 64, WMW = NWV / WNW, TM = WMW * 128;
  static constexpr int KP = TN / 64;                             // channel-block pairs per stage (1 | 2)
  static constexpr int NB = 3;                                   // ring depth
  static constexpr int WPIECES = KP * TN * 2, XPIECES = KP * TM * 2;
  static constexpr int WSLOTS = (WPIECES + NT - 1) / NT * NT;    // weight region: whole workgroup-instructions
  static constexpr int PW = WSLOTS / NT, PX = XPIECES / NT, LPS = PW + PX;
  static constexpr int BUF16 = WSLOTS + XPIECES;
  static constexpr int LDS_BYTES = NB * BUF16 * 16;
};

// MS: multi-source (concat / collage) input.  EK: epilogue kind, chosen by the launcher -- 0 = generic (fp32 outputs, fp32 gate /
// residual: op tests and the configurations on the fp32 attention core), 1 / 2 / 3 = conv1_epilogue_stream<EK> (a kernel per
// kind: with all four in one kernel the register allocation of the widest one spilled)
template <int TN, int NWV, bool MS, int EK>
__global__ __launch_bounds__(NWV * 64, 2) void conv1_bf16(ConvArgsH ah, const void* zero_page, ConcatX cx) {
  using G = H1Geo<TN, NWV>;
  constexpr int NT = G::NT;
  static_assert(G::WPIECES <= G::WSLOTS && G::XPIECES % NT == 0 && G::WSLOTS % NT == 0, "staging shape");
  const ConvArgs& a = ah.c;
  extern __shared__ __attribute__((aligned(16))) u32x4 lds16[];
#ifdef TM_STAMPS
  unsigned long long* stamp_slot = nullptr;
  if (threadIdx.x == 0 && g_tm_stamps) {
    const unsigned int k = atomicAdd(&g_tm_stamp_next, 1u);
    if (k < g_tm_stamp_cap) {
      stamp_slot = g_tm_stamps + (size_t)k * 16;
      stamp_slot[4] = gridDim.x; stamp_slot[5] = blockIdx.x;
      stamp_slot[6] = (unsigned long long)ah.Cbp * 1000000ull + 500000ull + (unsigned long long)(TN * 1000 + NWV * 10 + (MS ? 1 : 0));
      stamp_slot[7] = __builtin_amdgcn_s_memrealtime();
    }
  }
  TM_STAMP(0);
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int i32 = lane & 31, h = lane >> 5;
  const int wn = wv % G::WNW, wm = wv / G::WNW;
  const int bid = xcd_swizzle(blockIdx.x, gridDim.x);
  const int nt = bid % a.ntile;
  const int mtile = bid / a.ntile;
  const long VPN = (long)a.Z * a.S * a.S, vtot = VPN * a.N;
  const h16_t* xg = (const h16_t*)a.x;
  const h16_t* wg = (const h16_t*)a.w;
  const h16_t* zp = (const h16_t*)zero_page;
  // voxel -> (patch, in-patch index): the tile's first voxel is split by one wave-uniform 64-bit division, every lane's voxel
  // by a 32-bit one from there (a 64-bit division per piece and per output voxel tile made the prologue 3 k cycles, 7.6 k with
  // the concat input's index arithmetic on top)
  const long vg0 = (long)mtile * G::TM;
  const int VPNi = (int)VPN;
  const long n0 = vtot <= 0x7fffffffL ? (long)((int)vg0 / VPNi) : vg0 / VPN;
  const int rem0 = (int)(vg0 - n0 * VPN);
  const long vleft = vtot - vg0;                       // voxels from the tile's first one to the end of the tensor
  const bool dsm = vtot <= 0x7fffffffL && a.N < (1 << 21) && VPN + G::TM < (1 << 21);      // idiv_small's range
  const float rVPN = 1.0f / (float)VPNi;

  // piece i = tid + k*NT -> LDS slot WSLOTS + i ([kp][k-half][voxel]); (kp, half) are the same for every lane of an
  // instruction (NT divides TM or equals it), the voxel differs per lane.  Per piece: the voxel's element offset inside
  // a channel-block plane and its patch index, as they stand and (multi-source input) after the collage remap.
  long xoff[G::PX];
  int xkp[G::PX], xhalf[G::PX];
  // multi-source only: per source and piece, the voxel's element offset inside the source (patch stride x patch + in-plane
  // offset, through the collage remap if the source is collaged), -1 for a voxel past the end.  Computed here, once: in
  // the K loop a piece costs two selects, one 64-bit add and the scalar block term (the 64-bit multiplies per piece and
  // stage that stood there made a stage 3 000-3 200 cycles against 2 300-2 500 of the single-source kernel)
  long xbase[3][G::PX];
  constexpr bool multi = MS;
  const h16_t *ms_p0 = (const h16_t*)cx.p0, *ms_p1 = (const h16_t*)cx.p1p, *ms_p2 = (const h16_t*)cx.p2p;
  const long ms_ns0 = cx.ns0, ms_ns1 = cx.ns1, ms_ns2 = cx.ns2;
  const int ms_cb1 = cx.cb1, ms_cb2 = cx.cb2, ms_cbtot = cx.cbtot;
  const bool ms_col0 = cx.col0 != 0, ms_col1 = cx.col1 != 0, ms_col2 = cx.col2 != 0;
  const int ms_S = a.S, ms_q1 = cx.p1 - 1, ms_q2 = cx.p2 - 1, ms_pp1 = cx.p1, ms_pp2 = cx.p2;
  const long ms_plane = ah.x_plane_e;
  const float ms_rSS = 1.0f / (float)(ms_S * ms_S), ms_rS = 1.0f / (float)ms_S;
  const float ms_rq12 = 1.0f / (float)(ms_q1 * ms_q2), ms_rq2 = 1.0f / (float)ms_q2;
#pragma unroll
  for (int k = 0; k < G::PX; ++k) {
    const int i = tid + k * NT;
    const int v = i % G::TM;
    const int half = (i / G::TM) & 1;
    xkp[k] = i / (2 * G::TM);
    xhalf[k] = half;
    long off = -1;
    xbase[0][k] = -1; xbase[1][k] = -1; xbase[2][k] = -1;
    if (v < vleft) {
      const int r0 = rem0 + v, dn = idiv_small(r0, VPNi, rVPN, dsm);     // the tile's first voxel was split by ONE division
      const long n = n0 + dn;
      const int rem = r0 - dn * VPNi;
      off = n * ah.x_nstride_e + (long)half * ah.x_plane_e + (long)rem * 8;
      if (multi) {
        const int S = ms_S;
        const int z = idiv_small(rem, S * S, ms_rSS, dsm), r2 = rem - z * S * S;
        const int y = idiv_small(r2, S, ms_rS, dsm), x = r2 - y * S;
        const int q1 = ms_q1, q2 = ms_q2;
        const int bi = idiv_small((int)n, q1 * q2, ms_rq12, dsm), q = (int)n - bi * q1 * q2;
        int pi = idiv_small(q, q2, ms_rq2, dsm), pj = q - pi * q2;
        int ys = y + S / 2, xs = x + S / 2;
        if (ys >= S) { ys -= S; pi += 1; }
        if (xs >= S) { xs -= S; pj += 1; }
        const long nc = bi * ms_pp1 * ms_pp2 + pi * ms_pp2 + pj;
        const int oc = ((z * S + ys) * S + xs) * 8;
        xbase[0][k] = ms_col0 ? nc * ms_ns0 + oc : n * ms_ns0 + rem * 8;
        xbase[1][k] = ms_col1 ? nc * ms_ns1 + oc : n * ms_ns1 + rem * 8;
        xbase[2][k] = ms_col2 ? nc * ms_ns2 + oc : n * ms_ns2 + rem * 8;
      }
    }
    xoff[k] = off;
  }
  // packed weights: [n-tile][pair][TN][2][8]; piece i = tid + k*NT of a stage belongs to pair i / (TN*2)
  const h16_t* wsrc = wg + (long)nt * ah.Cbp * TN * 16 + (long)tid * 8;

  int xb[4], on[4], ooff[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int v = (wm * 4 + mt) * 32 + i32;
    xb[mt] = G::WSLOTS + h * G::TM + v;                // consecutive lanes, consecutive slots: conflict free
    if (v < vleft) {
      const int r0 = rem0 + v, dn = idiv_small(r0, VPNi, rVPN, dsm);
      on[mt] = (int)n0 + dn;
      ooff[mt] = (r0 - dn * VPNi) * 8;
    } else { on[mt] = 0; ooff[mt] = -1; }
  }
  const int wb = (wn * 64 + i32) * 2 + (h ^ ((i32 >> 3) & 1));

  f32x16 acc[2][4];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ct][mt][r] = 0.f;

  const int NS = (ah.Cbp + G::KP - 1) / G::KP;
  // One loop over it = -(NB-1) .. NS-1: iteration `it` first issues stage it + NB - 1 (exactly LPS LDS-DMA instructions,
  // whatever the stage is: the ring fill, the steady state and the dummy tail are the same code, at ONE site -- a
  // capturing lambda used from two places kept every captured local, the kernel arguments included, in scratch) and then,
  // for it >= 0, computes stage `it`.
  TM_STAMP(1);
  for (int it = -(G::NB - 1); it < NS; ++it) {
    if (it >= 0) {
      // stage `it` has landed once at most (NB-2) younger stages of this wave are still in flight
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((G::NB - 2) * G::LPS) : "memory");
      __builtin_amdgcn_s_barrier();                     // everyone's share landed; buffer (it-1)%NB is free
      __builtin_amdgcn_sched_barrier(0);
    }
    {
      const int st = it + G::NB - 1;
      const int p0 = st * G::KP;
      u32x4* base = lds16 + (st % G::NB) * G::BUF16;
      // the source address is selected arithmetically and laundered through a VGPR: a `cond ? ptr : zp` that
      // hipcc turns into two predicated DMA instructions would break the per-stage instruction count
#pragma unroll
      for (int k = 0; k < G::PW; ++k) {
        const int i = tid + k * NT;
        const bool wok = st < NS && i < G::WPIECES && p0 + i / (TN * 2) < ah.Cbp;
        unsigned long long wa = wok ? (unsigned long long)(wsrc + (long)p0 * TN * 16 + (long)k * NT * 8) : (unsigned long long)zp;
        asm volatile("" : "+v"(wa));
        TM_GLDS16((const void*)wa, base + k * NT + wv * 64);
      }
#pragma unroll
      for (int k = 0; k < G::PX; ++k) {
        const int pr = p0 + xkp[k];
        unsigned long long xa;
        if (multi) {
          const int vb = 2 * pr + xhalf[k];              // virtual channel block of the concat (wave-uniform)
          // source select by scalar compares (no dynamically indexed kernel-argument loads inside the K loop)
          const bool g1 = vb >= ms_cb1, g2 = vb >= ms_cb2;
          const h16_t* sp = g2 ? ms_p2 : (g1 ? ms_p1 : ms_p0);
          const int scb = g2 ? ms_cb2 : (g1 ? ms_cb1 : 0);
          const long xbk = g2 ? xbase[2][k] : (g1 ? xbase[1][k] : xbase[0][k]);
          const bool ok = st < NS && xbk >= 0 && vb < ms_cbtot;
          xa = ok ? (unsigned long long)(sp + (long)(vb - scb) * ms_plane + xbk) : (unsigned long long)zp;
        } else {
          const bool ok = st < NS && xoff[k] >= 0 && pr < ah.Cbp;
          xa = ok ? (unsigned long long)(xg + (long)pr * 2 * ah.x_plane_e + xoff[k]) : (unsigned long long)zp;
        }
        asm volatile("" : "+v"(xa));
        TM_GLDS16((const void*)xa, base + G::WSLOTS + k * NT + wv * 64);
      }
    }
    if (it < 0) continue;
    const u32x4* buf = lds16 + (it % G::NB) * G::BUF16;
#pragma unroll
    for (int kp = 0; kp < G::KP; ++kp) {
      bf16x8 wf[2], xf[4];
      wf[0] = __builtin_bit_cast(bf16x8, buf[kp * TN * 2 + wb]);
      wf[1] = __builtin_bit_cast(bf16x8, buf[kp * TN * 2 + 64 + wb]);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) xf[mt] = __builtin_bit_cast(bf16x8, buf[xb[mt] + kp * G::TM * 2]);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          acc[ct][mt] = TM_MFMA16(wf[ct], xf[mt], acc[ct][mt]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // drain the tail's dummy DMAs before the LDS is released
  TM_STAMP(2);
  const int cob0 = (nt * G::WNW + wn) * 8;
  if (EK == 0) conv_epilogue_h16<true>(a, acc, cob0, h, on, ooff);
  else if (EK == 3) conv1_epilogue_gated(a, acc, cob0, h, on, ooff);
  else conv1_epilogue_stream<EK>(a, acc, cob0, h, on, ooff);
#ifdef TM_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TM_STAMP(3);
#endif
}


// ==========================================================================================
// Windowed cross attention core, bf16 operands (TM_DTYPE_BF16): model/MBAblocks.py:560-590 with
// enable_flash_attn=True semantics -- q_norm / k_norm (RMSNorm over C), softmax((q*scale).(k*scale)^T), .v,
// scale = C^-1/2, one head of width C, windows of T = Z*(S/2)^2 tokens (2 x 2 windows per patch).
// q, k, v arrive as bf16 CB8 (written by the q / kv Linears), the result leaves as bf16 CB8 (proj's input).
//
//   S^T = K.Q^T  on v_mfma_f32_32x32x16_bf16: A = K (rows = keys) read as raw 16-byte fragments straight from
//        global memory, B = Q (cols = queries) with q_norm.w * k_norm.w folded in; the two rstd factors and the
//        scale are applied to the fp32 result.  A lane then owns ONE query and 16 x T/32 keys: the softmax
//        reduction is in-lane plus one exchange with lane ^ 32.
//   P -> LDS row-major [query][key] bf16 (8-byte stores of 4 consecutive keys), which is the B operand of
//   O^T = V^T.P^T: A = V^T (rows = channels) from an LDS image transposed while staging, 64 channels per
//        stage, double buffered.  A lane ends up with one token and 4 consecutive channels per accumulator quad:
//        the coalesced bf16 CB8 store of the conv epilogues.
// T = 128: one workgroup per window, wave w owns queries [32w, 32w+32).  T = 64: two windows per workgroup, two waves
// each.  T = 32: one workgroup per patch, one window per wave.  A workgroup always covers 128 query and 128 key tokens.
template <int T>
struct WAGeo {
  static constexpr int NW = T / 32;                 // waves per window
  static constexpr int WPW = 4 / NW;                // windows per workgroup
  static constexpr int PP = T * 2 + 16;             // row pitch in bytes of P and V^T rows: 16 B mod 256 B
  static constexpr int CH = 64;                     // channels per V stage
  static constexpr int P_BYTES = 128 * PP;          // [128 queries of the workgroup][T keys]
  static constexpr int VT_BYTES = WPW * CH * PP;    // one stage, all windows of the workgroup
  static constexpr int MISC_FLOATS = 128 /*tokoff*/ + 128 /*tokoff of k / v*/ + 128 /*rq*/ + 128 /*rk*/ + 512 /*w2*/;
  static constexpr int LDS_BYTES = P_BYTES + 2 * VT_BYTES + MISC_FLOATS * 4;
};

struct WinArgsH {
  const uint16_t *q, *k, *v; long q_ns, k_ns, v_ns;
  const float *qw, *kw;
  uint16_t* o; long o_ns;
  int C, S;
  long plane;                                       // elements per channel block
  int kv_half;                                      // k / v live at half the in-plane resolution (token (z, y, x) reads (z, y >> 1, x >> 1))
};

template <int T>
__global__ __launch_bounds__(256, 2) void window_attn_bf16(WinArgsH a) {
  using G = WAGeo<T>;
  typedef h16_t bf16x4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Pl = smem;                                   // [128][PP]
  unsigned char* Vt = Pl + G::P_BYTES;                        // 2 x [WPW][CH][PP]
  int* tokoff = (int*)(Vt + 2 * G::VT_BYTES);                 // [128]: window-major, token-minor
  int* tokoff_kv = tokoff + 128;                              // the same tokens in the k / v tensors (half resolution or equal)
  float* rq = (float*)(tokoff_kv + 128);
  float* rk = rq + 128;
  float* w2 = rk + 128;                                       // [C] q_norm.w * k_norm.w
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int i32 = lane & 31, h = lane >> 5;
  constexpr int NWG = 4 / G::WPW;                            // workgroups per patch (each covers WPW windows = 128 tokens)
  const int n = blockIdx.x / NWG;
  const int S = a.S, hs = S / 2, C = a.C, npair = C / 16;
  const long kv_plane = a.kv_half ? a.plane >> 2 : a.plane;
  if (tid < 128) {
    const int win = (blockIdx.x % NWG) * G::WPW + tid / T;
    const int t = tid % T;
    const int wy = win >> 1, wx = win & 1;
    const int z = t / (hs * hs);
    const int r = t - z * hs * hs;
    const int yl = r / hs, xl = r - yl * hs;
    const int yy = wy * hs + yl, xx = wx * hs + xl;
    tokoff[tid] = ((z * S + yy) * S + xx) * 8;
    tokoff_kv[tid] = a.kv_half ? ((z * hs + (yy >> 1)) * hs + (xx >> 1)) * 8 : ((z * S + yy) * S + xx) * 8;
  }
  for (int c = tid; c < C; c += 256) w2[c] = a.qw[c] * a.kw[c];
  __syncthreads();
  const h16_t* qb = (const h16_t*)a.q + (long)n * a.q_ns;
  const h16_t* kb = (const h16_t*)a.k + (long)n * a.k_ns;
  const h16_t* vb = (const h16_t*)a.v + (long)n * a.v_ns;
  {  // RMSNorm statistics of the 128 query and 128 key tokens (fp32 sums over the bf16 values)
    const bool isq = tid < 128;
    const int t = tid & 127;
    const h16_t* p = isq ? qb + tokoff[t] : kb + tokoff_kv[t];
    const long pp = isq ? a.plane : kv_plane;
    float ss = 0.f;
#pragma unroll 8                                              // eight independent 16-byte loads in flight (same summation order)
    for (int cb = 0; cb < C / 8; ++cb) {
      const bf16x8 v8 = *(const bf16x8*)(p + (long)cb * pp);
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float f = (float)v8[j]; ss += f * f; }
    }
    const float r = 1.0f / sqrtf(ss / (float)C + TM_EPS);
    if (isq) rq[t] = r; else rk[t] = r;
  }
  __syncthreads();

  const int wbase = (wv / G::NW) * T;                         // first token (workgroup numbering) of this wave's window
  const int qtok = wv * 32 + i32;                             // this lane's query (= wbase + 32 * (wv % NW) + i32)
  // ---- S^T = K.Q^T ----
  f32x16 acc[G::NW];
#pragma unroll
  for (int ct = 0; ct < G::NW; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
  const h16_t* qp = qb + tokoff[qtok] + (long)h * a.plane;
  const h16_t* kp[G::NW];
#pragma unroll
  for (int ct = 0; ct < G::NW; ++ct) kp[ct] = kb + tokoff_kv[wbase + ct * 32 + i32] + (long)h * kv_plane;
  // The operands come straight from global memory (each lane its own token's 16-byte entries): with one channel pair
  // prefetched ahead, every one of the C / 16 steps waited for a load issued four MFMAs earlier -- the loop was a chain of
  // memory round trips.  PD pairs are in flight now (C / 16 is a multiple of 4 for every width of the model family; otherwise
  // the one-ahead loop below).
  constexpr int PD = 4;
  if (npair % PD == 0) {
    bf16x8 qs[PD], ks[PD][G::NW];
#pragma unroll
    for (int u = 0; u < PD; ++u) {
      qs[u] = *(const bf16x8*)(qp + (long)u * 2 * a.plane);
#pragma unroll
      for (int ct = 0; ct < G::NW; ++ct) ks[u][ct] = *(const bf16x8*)(kp[ct] + (long)u * 2 * kv_plane);
    }
    for (int kp0 = 0; kp0 < npair; kp0 += PD) {
#pragma unroll
      for (int u = 0; u < PD; ++u) {
        const int kp2 = kp0 + u;
        bf16x8 qf;
        const bf16x8 qc = qs[u];
        bf16x8 kf[G::NW];
#pragma unroll
        for (int ct = 0; ct < G::NW; ++ct) kf[ct] = ks[u][ct];
        if (kp2 + PD < npair) {
          const long po = (long)(kp2 + PD) * 2 * a.plane, pk = (long)(kp2 + PD) * 2 * kv_plane;
          qs[u] = *(const bf16x8*)(qp + po);
#pragma unroll
          for (int ct = 0; ct < G::NW; ++ct) ks[u][ct] = *(const bf16x8*)(kp[ct] + pk);
        }
        {
          const f32x4 wa = *(const f32x4*)(w2 + kp2 * 16 + 8 * h), wb = *(const f32x4*)(w2 + kp2 * 16 + 8 * h + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { qf[j] = (h16_t)((float)qc[j] * wa[j]); qf[4 + j] = (h16_t)((float)qc[4 + j] * wb[j]); }
        }
#pragma unroll
        for (int ct = 0; ct < G::NW; ++ct) acc[ct] = TM_MFMA16(kf[ct], qf, acc[ct]);
      }
    }
  } else {
  bf16x8 qn = *(const bf16x8*)qp, kn[G::NW];
#pragma unroll
  for (int ct = 0; ct < G::NW; ++ct) kn[ct] = *(const bf16x8*)kp[ct];
  for (int kp2 = 0; kp2 < npair; ++kp2) {
    bf16x8 qf;
    {
      const f32x4 wa = *(const f32x4*)(w2 + kp2 * 16 + 8 * h), wb = *(const f32x4*)(w2 + kp2 * 16 + 8 * h + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { qf[j] = (h16_t)((float)qn[j] * wa[j]); qf[4 + j] = (h16_t)((float)qn[4 + j] * wb[j]); }
    }
    bf16x8 kf[G::NW];
#pragma unroll
    for (int ct = 0; ct < G::NW; ++ct) kf[ct] = kn[ct];
    if (kp2 + 1 < npair) {
      const long po = (long)(kp2 + 1) * 2 * a.plane, pk = (long)(kp2 + 1) * 2 * kv_plane;
      qn = *(const bf16x8*)(qp + po);
#pragma unroll
      for (int ct = 0; ct < G::NW; ++ct) kn[ct] = *(const bf16x8*)(kp[ct] + pk);
    }
#pragma unroll
    for (int ct = 0; ct < G::NW; ++ct) acc[ct] = TM_MFMA16(kf[ct], qf, acc[ct]);
  }
  }
  // ---- scale + softmax over the keys of this lane's query (registers, then lane ^ 32) ----
  {
    const float sq = rq[qtok] / (float)C;                     // (q*scale).(k*scale), scale = C^-1/2 (MBAblocks.py:571-577)
    float m = -INFINITY;
#pragma unroll
    for (int ct = 0; ct < G::NW; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        acc[ct][r] *= sq * rk[wbase + key];
        m = fmaxf(m, acc[ct][r]);
      }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float ssum = 0.f;
#pragma unroll
    for (int ct = 0; ct < G::NW; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[ct][r] = expf(acc[ct][r] - m); ssum += acc[ct][r]; }
    ssum += __shfl_xor(ssum, 32, 64);
    const float inv = 1.0f / ssum;
    unsigned char* prow = Pl + (long)(wv * 32 + i32) * G::PP;
#pragma unroll
    for (int ct = 0; ct < G::NW; ++ct)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 pk;
#pragma unroll
        for (int j = 0; j < 4; ++j) pk[j] = (h16_t)(acc[ct][4 * g + j] * inv);
        *(bf16x4*)(prow + (ct * 32 + 8 * g + 4 * h) * 2) = pk;
      }
  }
  // ---- O^T = V^T.P^T ----
  // V staging: 512 items per stage = 64 token pairs x 8 channel blocks; thread owns items tid and tid + 256
  const int gp = tid & 63;                                    // token pair (workgroup numbering 2gp, 2gp+1)
  const int swin = (2 * gp) / T, st = (2 * gp) % T;
  const h16_t* vs0 = vb + tokoff_kv[2 * gp];
  const h16_t* vs1 = vb + tokoff_kv[2 * gp + 1];
  const int scb = tid >> 6;                                   // channel blocks scb and scb + 4 of the stage
  bf16x8 vr[4];
  auto vload = [&](int c0) {
    const long o0 = (long)(c0 / 8 + scb) * kv_plane, o1 = o0 + 4 * kv_plane;
    vr[0] = *(const bf16x8*)(vs0 + o0); vr[1] = *(const bf16x8*)(vs1 + o0);
    vr[2] = *(const bf16x8*)(vs0 + o1); vr[3] = *(const bf16x8*)(vs1 + o1);
  };
  auto vstore = [&](int buf) {
    unsigned char* base = Vt + buf * G::VT_BYTES + swin * G::CH * G::PP + st * 2;
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        typedef h16_t bf16x2 __attribute__((ext_vector_type(2)));
        bf16x2 two;
        two[0] = vr[2 * half][j]; two[1] = vr[2 * half + 1][j];
        *(bf16x2*)(base + ((scb + 4 * half) * 8 + j) * G::PP) = two;
      }
  };
  vload(0);
  vstore(0);
  if (G::CH < C) vload(G::CH);
  __syncthreads();                                            // P rows and V stage 0 visible
  bf16x8 pf[T / 16];
  {
    const unsigned char* prow = Pl + (long)(wv * 32 + i32) * G::PP + 16 * h;
#pragma unroll
    for (int kb2 = 0; kb2 < T / 16; ++kb2) pf[kb2] = *(const bf16x8*)(prow + kb2 * 32);
  }
  const int myoff = tokoff[qtok];
  const int vwin = wv / G::NW;
  h16_t* ob = (h16_t*)a.o + (long)n * a.o_ns;
  int buf = 0;
  for (int c0 = 0; c0 < C; c0 += G::CH) {
    if (c0 + G::CH < C) vstore(buf ^ 1);                      // next stage's registers -> the other buffer
    if (c0 + 2 * G::CH < C) vload(c0 + 2 * G::CH);
    const unsigned char* vrow = Vt + buf * G::VT_BYTES + vwin * G::CH * G::PP + 16 * h;
#pragma unroll
    for (int ctile = 0; ctile < G::CH / 32; ++ctile) {
      f32x16 oc;
#pragma unroll
      for (int r = 0; r < 16; ++r) oc[r] = 0.f;
      const unsigned char* ar = vrow + (long)(ctile * 32 + i32) * G::PP;
#pragma unroll
      for (int kb2 = 0; kb2 < T / 16; ++kb2)
        oc = TM_MFMA16(*(const bf16x8*)(ar + kb2 * 32), pf[kb2], oc);
      // lane = token qtok; accumulator quad g = channels c0 + 32*ctile + 8g + 4h .. +3
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 o4;
#pragma unroll
        for (int j = 0; j < 4; ++j) o4[j] = (h16_t)oc[4 * g + j];
        *(bf16x4*)(ob + myoff + (long)((c0 + 32 * ctile) / 8 + g) * a.plane + 4 * h) = o4;
      }
    }
    __syncthreads();                                          // stage `buf` consumed, stage buf^1 complete
    buf ^= 1;
  }
}


// ---- long windows (T = 256 / 512: the z_size 4 / 8 models at the resolution-16 attention blocks) --------------------
// Same operand scheme as window_attn_bf16, but the keys are walked in blocks of 128 with an exact two-pass softmax:
// pass 1 accumulates every query's running max / sum over all key blocks (S^T = K.Q^T only), pass 2 recomputes each
// block's logits, writes P = exp(s - m) / l for that block to LDS and adds V_blk^T.P_blk^T into the output accumulators
// (C / 32 tiles per wave, kept across key blocks).  One workgroup = 128 queries of one window; C <= 256.
template <int T>
struct WLGeo {
  static constexpr int QBN = T / 128;                // query blocks (workgroups) per window
  static constexpr int KBN = T / 128;                // key blocks
  static constexpr int PP = 128 * 2 + 16;            // row pitch (bytes) of the P block and of a V^T row
  static constexpr int P_BYTES = 128 * PP, VT_BYTES = 64 * PP;
  static constexpr int LDS_BYTES = P_BYTES + VT_BYTES + (T /*tokoff*/ + 128 /*rq*/ + T /*rk*/ + 256 /*w2*/) * 4;
};

template <int T>
__global__ __launch_bounds__(256, 1) void window_attn_long(WinArgsH a) {
  using G = WLGeo<T>;
  typedef h16_t bf16x4 __attribute__((ext_vector_type(4)));
  typedef h16_t bf16x2 __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Pl = smem;                                   // [128 queries][128 keys of the current block]
  unsigned char* Vt = Pl + G::P_BYTES;                        // [64 channels][128 keys]
  int* tokoff = (int*)(Vt + G::VT_BYTES);                     // [T]
  float* rq = (float*)(tokoff + T);                           // [128]
  float* rk = rq + 128;                                       // [T]
  float* w2 = rk + T;                                         // [C]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int i32 = lane & 31, h = lane >> 5;
  const int qblk = blockIdx.x % G::QBN, win = (blockIdx.x / G::QBN) & 3, n = blockIdx.x / (4 * G::QBN);
  const int wy = win >> 1, wx = win & 1;
  const int S = a.S, hs = S / 2, C = a.C, npair = C / 16;
  for (int t = tid; t < T; t += 256) {
    const int z = t / (hs * hs);
    const int r = t - z * hs * hs;
    const int yl = r / hs, xl = r - yl * hs;
    tokoff[t] = ((z * S + wy * hs + yl) * S + wx * hs + xl) * 8;
  }
  for (int c = tid; c < C; c += 256) w2[c] = a.qw[c] * a.kw[c];
  __syncthreads();
  const h16_t* qb = (const h16_t*)a.q + (long)n * a.q_ns;
  const h16_t* kb = (const h16_t*)a.k + (long)n * a.k_ns;
  const h16_t* vb = (const h16_t*)a.v + (long)n * a.v_ns;
  for (int i = tid; i < 128 + T; i += 256) {                  // RMSNorm statistics: this block's queries, all keys
    const bool isq = i < 128;
    const int t = isq ? qblk * 128 + i : i - 128;
    const h16_t* p = (isq ? qb : kb) + tokoff[t];
    float ss = 0.f;
    for (int cb = 0; cb < C / 8; ++cb) {
      const bf16x8 v8 = *(const bf16x8*)(p + (long)cb * a.plane);
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float f = (float)v8[j]; ss += f * f; }
    }
    const float r = 1.0f / sqrtf(ss / (float)C + TM_EPS);
    if (isq) rq[i] = r; else rk[t] = r;
  }
  __syncthreads();

  const int qloc = wv * 32 + i32;                             // this lane's query inside the block
  const int qoff = tokoff[qblk * 128 + qloc];
  const float sq = rq[qloc] / (float)C;                       // (q*scale).(k*scale), scale = C^-1/2
  const h16_t* qp = qb + qoff + (long)h * a.plane;
  f32x16 acc[4];
  // logits of key block kblk against this wave's 32 queries: acc[ct][r] = s(key kblk*128 + ct*32 + row(r, h), query i32)
  auto s_block = [&](int kblk) {
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
    const h16_t* kp[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) kp[ct] = kb + tokoff[kblk * 128 + ct * 32 + i32] + (long)h * a.plane;
    bf16x8 qn = *(const bf16x8*)qp, kn[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) kn[ct] = *(const bf16x8*)kp[ct];
    for (int kp2 = 0; kp2 < npair; ++kp2) {
      bf16x8 qf, kf[4];
      const f32x4 wa = *(const f32x4*)(w2 + kp2 * 16 + 8 * h), wb = *(const f32x4*)(w2 + kp2 * 16 + 8 * h + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { qf[j] = (h16_t)((float)qn[j] * wa[j]); qf[4 + j] = (h16_t)((float)qn[4 + j] * wb[j]); }
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) kf[ct] = kn[ct];
      if (kp2 + 1 < npair) {
        const long po = (long)(kp2 + 1) * 2 * a.plane;
        qn = *(const bf16x8*)(qp + po);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) kn[ct] = *(const bf16x8*)(kp[ct] + po);
      }
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) acc[ct] = TM_MFMA16(kf[ct], qf, acc[ct]);
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ct][r] *= sq * rk[kblk * 128 + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
  };
  // ---- pass 1: running max / sum of this lane's query over all keys ----
  float m = -INFINITY, l = 0.f;
  for (int kblk = 0; kblk < G::KBN; ++kblk) {
    s_block(kblk);
    float bm = -INFINITY;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) bm = fmaxf(bm, acc[ct][r]);
    bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
    const float mn = fmaxf(m, bm);
    float bs = 0.f;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) bs += expf(acc[ct][r] - mn);
    bs += __shfl_xor(bs, 32, 64);
    l = l * expf(m - mn) + bs;
    m = mn;
  }
  const float inv = 1.0f / l;
  // ---- pass 2: P block by block, O^T += V_blk^T . P_blk^T ----
  f32x16 oc[8];
#pragma unroll
  for (int tI = 0; tI < 8; ++tI)
#pragma unroll
    for (int r = 0; r < 16; ++r) oc[tI][r] = 0.f;
  const int gp = tid & 63, scb = tid >> 6;                    // V staging: token pair gp, channel blocks scb and scb + 4
  for (int kblk = 0; kblk < G::KBN; ++kblk) {
    s_block(kblk);
    unsigned char* prow = Pl + (long)qloc * G::PP;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 pk;
#pragma unroll
        for (int j = 0; j < 4; ++j) pk[j] = (h16_t)(expf(acc[ct][4 * g + j] - m) * inv);
        *(bf16x4*)(prow + (ct * 32 + 8 * g + 4 * h) * 2) = pk;
      }
    const h16_t* vs0 = vb + tokoff[kblk * 128 + 2 * gp];
    const h16_t* vs1 = vb + tokoff[kblk * 128 + 2 * gp + 1];
    bf16x8 pf[8];
    for (int c0 = 0; c0 < C; c0 += 64) {
      __syncthreads();                                        // V^T buffer free; (first time) every wave's P rows written
      {
        const long o0 = (long)(c0 / 8 + scb) * a.plane, o1 = o0 + 4 * a.plane;
        const bf16x8 v00 = *(const bf16x8*)(vs0 + o0), v01 = *(const bf16x8*)(vs1 + o0);
        const bf16x8 v10 = *(const bf16x8*)(vs0 + o1), v11 = *(const bf16x8*)(vs1 + o1);
        unsigned char* base = Vt + gp * 4;                    // two keys = 4 bytes
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          bf16x2 t0, t1;
          t0[0] = v00[j]; t0[1] = v01[j]; t1[0] = v10[j]; t1[1] = v11[j];
          *(bf16x2*)(base + (scb * 8 + j) * G::PP) = t0;
          *(bf16x2*)(base + ((scb + 4) * 8 + j) * G::PP) = t1;
        }
      }
      __syncthreads();
      if (c0 == 0) {
        const unsigned char* pr = Pl + (long)qloc * G::PP + 16 * h;
#pragma unroll
        for (int kb2 = 0; kb2 < 8; ++kb2) pf[kb2] = *(const bf16x8*)(pr + kb2 * 32);
      }
      const unsigned char* vrow = Vt + 16 * h;
#pragma unroll
      for (int ctile = 0; ctile < 2; ++ctile) {
        const unsigned char* ar = vrow + (long)(ctile * 32 + i32) * G::PP;
        const int tI = c0 / 32 + ctile;
        // static register indexing: the c0 loop has at most 4 iterations (C <= 256)
#pragma unroll
        for (int cc = 0; cc < 8; ++cc)
          if (cc == tI) {
#pragma unroll
            for (int kb2 = 0; kb2 < 8; ++kb2) oc[cc] = TM_MFMA16(*(const bf16x8*)(ar + kb2 * 32), pf[kb2], oc[cc]);
          }
      }
    }
    __syncthreads();                                          // P block consumed before the next block overwrites it
  }
  h16_t* ob = (h16_t*)a.o + (long)n * a.o_ns;
#pragma unroll
  for (int tI = 0; tI < 8; ++tI) {
    if (tI * 32 >= C) break;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 o4;
#pragma unroll
      for (int j = 0; j < 4; ++j) o4[j] = (h16_t)oc[tI][4 * g + j];
      *(bf16x4*)(ob + qoff + (long)(tI * 4 + g) * a.plane + 4 * h) = o4;
    }
  }
}

hipError_t launch_window_attn_bf16(const TVH& q, const TVH& k, const TVH& v, const float* qnorm_w, const float* knorm_w,
                                   TVH o, hipStream_t s) {
  WinArgsH a;
  a.q = q.p; a.k = k.p; a.v = v.p; a.q_ns = q.nstride; a.k_ns = k.nstride; a.v_ns = v.nstride;
  a.qw = qnorm_w; a.kw = knorm_w; a.o = o.p; a.o_ns = o.nstride;
  a.C = q.Cb * 8; a.S = q.H; a.plane = (long)q.Z * q.H * q.W * 8;
  const int T = q.Z * (q.H / 2) * (q.H / 2);
  if (a.C % 64 || a.C > 512 || q.H != q.W || (q.H & 1)) return hipErrorInvalidValue;
  // k / v at half the in-plane resolution of q (the conditioning side of an AttnBlock is constant over 2 x 2 voxel blocks)
  a.kv_half = (k.H * 2 == q.H) ? 1 : 0;
  if ((!a.kv_half && k.H != q.H) || k.H != v.H || k.W != k.H || v.W != v.H) return hipErrorInvalidValue;
  if (a.kv_half && ((q.H & 3) || (T != 128 && T != 64 && T != 32))) return hipErrorInvalidValue;
  if (T == 256 || T == 512) {
    if (a.C > 256) return hipErrorInvalidValue;
#define TM_LAUNCHWL(T_)                                                                                     \
  do {                                                                                                      \
    static DevOnce attr_set;                                                                           \
    if (attr_set.need()) {                                                                                        \
      hipError_t e = hipFuncSetAttribute((const void*)window_attn_long<T_>,                                 \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, WLGeo<T_>::LDS_BYTES); \
      if (e != hipSuccess) return e;                                                                        \
      attr_set.mark();                                                                                      \
    }                                                                                                       \
    hipLaunchKernelGGL(window_attn_long<T_>, dim3((unsigned)(q.N * 4 * WLGeo<T_>::QBN)), dim3(256),         \
                       WLGeo<T_>::LDS_BYTES, s, a);                                                         \
  } while (0)
    if (T == 256) TM_LAUNCHWL(256); else TM_LAUNCHWL(512);
#undef TM_LAUNCHWL
    return hipGetLastError();
  }
  if (T != 128 && T != 64 && T != 32) return hipErrorInvalidValue;
#define TM_LAUNCHWA(T_)                                                                                     \
  do {                                                                                                      \
    static DevOnce attr_set;                                                                           \
    if (attr_set.need()) {                                                                                        \
      hipError_t e = hipFuncSetAttribute((const void*)window_attn_bf16<T_>,                                 \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, WAGeo<T_>::LDS_BYTES); \
      if (e != hipSuccess) return e;                                                                        \
      attr_set.mark();                                                                                      \
    }                                                                                                       \
    hipLaunchKernelGGL(window_attn_bf16<T_>, dim3((unsigned)(q.N * (4 / WAGeo<T_>::WPW))), dim3(256),       \
                       WAGeo<T_>::LDS_BYTES, s, a);                                                         \
  } while (0)
  if (T == 128) TM_LAUNCHWA(128); else if (T == 64) TM_LAUNCHWA(64); else TM_LAUNCHWA(32);
#undef TM_LAUNCHWA
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
#ifdef TM_H16_F16
static inline uint16_t f32_to_bf16_rne(float f) {      // here: fp32 -> IEEE half, round to nearest even (saturating to inf like the cast)
  const _Float16 hf = (_Float16)f;
  uint16_t u;
  memcpy(&u, &hf, 2);
  return u;
}
#else
static inline uint16_t f32_to_bf16_rne(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7F800000u) == 0x7F800000u && (u & 0x007FFFFFu)) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
#endif

#ifndef TM_H16_F16
int conv_bf16_tn(int Cout) { return Cout <= 64 ? 64 : 128; }
#endif

#ifndef TM_H16_F16
size_t conv_bf16_pack_elems(int Cout, int Cbi) {
  const int TN = conv_bf16_tn(Cout);
  const int ntile = (Cout + TN - 1) / TN, Cbp = (Cbi + 1) / 2;
  return (size_t)ntile * Cbp * 27 * TN * 16;
}
#endif

void conv_bf16_pack_host(const float* w, int Cout, const int* seg_c, int nseg, uint16_t* out) {
  int Cin = 0, Cbi = 0;
  for (int s = 0; s < nseg; ++s) { Cin += seg_c[s]; Cbi += (seg_c[s] + 7) / 8; }
  const int TN = conv_bf16_tn(Cout), Cbp = (Cbi + 1) / 2;
  memset(out, 0, conv_bf16_pack_elems(Cout, Cbi) * sizeof(uint16_t));
  int ci0 = 0, cb0 = 0;
  for (int s = 0; s < nseg; ++s) {
    for (int c = 0; c < seg_c[s]; ++c) {
      const int ci = ci0 + c, cb = cb0 + c / 8, c8 = c % 8;
      const int pair = cb >> 1, half = cb & 1;
      for (int co = 0; co < Cout; ++co) {
        const int nt = co / TN, col = co % TN;
        const int slot = half ^ ((col >> 3) & 1);                 // bank-conflict-free weight fragment reads
        const float* src = w + ((size_t)co * Cin + ci) * 27;
        uint16_t* dst = out + ((((size_t)nt * Cbp + pair) * 27) * TN + col) * 16 + slot * 8 + c8;
        for (int t = 0; t < 27; ++t) dst[(size_t)t * TN * 16] = f32_to_bf16_rne(src[t]);
      }
    }
    ci0 += seg_c[s];
    cb0 += (seg_c[s] + 7) / 8;
  }
}

// Phase weights of the upsampled-input conv (HGeo UPS; the fp32 twin is conv_pack_ups_host): per output phase (py, px) the
// 3 x 3 in-plane taps collapse onto a 2 x 2 window of the low-resolution input -- rows: py = 0: {ky 0} | {ky 1, 2};
// py = 1: {ky 0, 1} | {ky 2}; columns alike; the sums are formed in fp32 and rounded once.
// out: [phase = 2 py + px][n-tile][pair][kz 3][4 taps (wy, wx)][TN][2][8]
#ifndef TM_H16_F16
size_t conv_bf16_pack_ups_elems(int Cout, int Cbi) {
  const int TN = conv_bf16_tn(Cout);
  return (size_t)4 * ((Cout + TN - 1) / TN) * ((Cbi + 1) / 2) * 12 * TN * 16;
}
#endif
void conv_bf16_pack_ups_host(const float* w, int Cout, const int* seg_c, int nseg, uint16_t* out) {
  int Cin = 0, Cbi = 0;
  for (int s = 0; s < nseg; ++s) { Cin += seg_c[s]; Cbi += (seg_c[s] + 7) / 8; }
  const int TN = conv_bf16_tn(Cout), Cbp = (Cbi + 1) / 2, ntile = (Cout + TN - 1) / TN;
  static const int G0[2][2][2] = {{{0, 0}, {1, 2}}, {{0, 1}, {2, 2}}};       // [phase bit][window pos] -> tap range [lo, hi]
  memset(out, 0, conv_bf16_pack_ups_elems(Cout, Cbi) * sizeof(uint16_t));
  for (int ph = 0; ph < 4; ++ph) {
    const int py = ph >> 1, px = ph & 1;
    int ci0 = 0, cb0 = 0;
    for (int s = 0; s < nseg; ++s) {
      for (int c = 0; c < seg_c[s]; ++c) {
        const int ci = ci0 + c, cb = cb0 + c / 8, c8 = c % 8;
        const int pair = cb >> 1, half = cb & 1;
        for (int co = 0; co < Cout; ++co) {
          const int nt = co / TN, col = co % TN;
          const int slot = half ^ ((col >> 3) & 1);
          const float* src = w + ((size_t)co * Cin + ci) * 27;
          uint16_t* dst = out + (((((size_t)ph * ntile + nt) * Cbp + pair) * 12) * TN + col) * 16 + slot * 8 + c8;
          for (int kz = 0; kz < 3; ++kz)
            for (int wy = 0; wy < 2; ++wy)
              for (int wx = 0; wx < 2; ++wx) {
                float acc = 0.f;
                for (int ky = G0[py][wy][0]; ky <= G0[py][wy][1]; ++ky)
                  for (int kx = G0[px][wx][0]; kx <= G0[px][wx][1]; ++kx) acc += src[kz * 9 + ky * 3 + kx];
                dst[(size_t)(kz * 4 + wy * 2 + wx) * TN * 16] = f32_to_bf16_rne(acc);
              }
        }
      }
      ci0 += seg_c[s];
      cb0 += (seg_c[s] + 7) / 8;
    }
  }
}

#ifndef TM_H16_F16
size_t conv1_bf16_pack_elems(int Cout, int Cbi) {
  const int TN = conv_bf16_tn(Cout);
  return (size_t)((Cout + TN - 1) / TN) * ((Cbi + 1) / 2) * TN * 16;
}
#endif
void conv1_bf16_pack_host(const float* w /*[Cout][Cin]*/, int Cout, const int* seg_c, int nseg, uint16_t* out) {
  int Cin = 0, Cbi = 0;
  for (int s = 0; s < nseg; ++s) { Cin += seg_c[s]; Cbi += (seg_c[s] + 7) / 8; }
  const int TN = conv_bf16_tn(Cout), Cbp = (Cbi + 1) / 2;
  memset(out, 0, conv1_bf16_pack_elems(Cout, Cbi) * sizeof(uint16_t));
  int ci0 = 0, cb0 = 0;
  for (int s = 0; s < nseg; ++s) {
    for (int c = 0; c < seg_c[s]; ++c) {
      const int ci = ci0 + c, cb = cb0 + c / 8, c8 = c % 8;
      const int pair = cb >> 1, half = cb & 1;
      for (int co = 0; co < Cout; ++co) {
        const int nt = co / TN, col = co % TN;
        const int slot = half ^ ((col >> 3) & 1);
        out[(((size_t)nt * Cbp + pair) * TN + col) * 16 + slot * 8 + c8] = f32_to_bf16_rne(w[(size_t)co * Cin + ci]);
      }
    }
    ci0 += seg_c[s];
    cb0 += (seg_c[s] + 7) / 8;
  }
}

static const void* zero_page(hipError_t* err) {
  static std::atomic<void*> zp[128];                    // 256 B of zeros per device, lives for the process
  const int d = DevOnce::dev();
  if (d < 0 || d >= 128) { *err = hipErrorInvalidDevice; return nullptr; }
  if (!zp[d].load()) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, 256);
    if (e == hipSuccess) e = hipMemset(p, 0, 256);
    if (e != hipSuccess) { *err = e; return nullptr; }
    void* expect = nullptr;
    if (!zp[d].compare_exchange_strong(expect, p)) (void)hipFree(p);       // another thread was first
  }
  return zp[d].load();
}
#ifdef TM_H16_F16
hipError_t init_f16_device() {
#else
hipError_t init_bf16_device() {
#endif
  // everything a first launch would otherwise set up lazily that is NOT a stream operation (a device allocation and a
  // synchronous memset): done at tm_model_finalize so that a forward can be captured into a graph from its first call
  hipError_t e = hipSuccess;
  return zero_page(&e) ? hipSuccess : e;
}

// A/B switches (environment, read once): TM_CONV1_WAVES / TM_CONV27_WAVES = 4 | 8 force the workgroup form of every
// launch that does not name one itself (ConvLaunchH::force_waves, the tm_op_* test entry points)
static int env_waves(const char* name) {
  const char* e = getenv(name);
  const int v = e ? atoi(e) : 0;
  return (v == 4 || v == 8) ? v : 0;
}

hipError_t launch_conv1_bf16(const ConvLaunchH& L, hipStream_t s) {
  hipError_t zerr = hipSuccess;
  const void* zp = zero_page(&zerr);
  if (!zp) return zerr;
  ConvArgsH ah;
  ConvArgs& a = ah.c;
  a.x = (const float*)L.x.p; a.x_nstride = 0; a.x_plane = 0;
  a.w = (const float*)L.w; a.bias = L.bias;
  a.y = L.y.p; a.y_nstride = L.y.nstride; a.y_plane = L.y.plane(); a.Cob = L.y.Cb;
  a.res = L.res ? L.res->p : nullptr; a.res_nstride = L.res ? L.res->nstride : 0;
  a.gate = L.gate ? L.gate->p : nullptr; a.gate_nstride = L.gate ? L.gate->nstride : 0;
  a.gate_h = L.gate_h ? L.gate_h->p : nullptr; a.gate_h_nstride = L.gate_h ? L.gate_h->nstride : 0;
  if (L.gate_half) {
    int ls = 0;
    while ((1 << ls) < L.y.H) ++ls;
    if (!L.gate_h || (1 << ls) != L.y.H || L.y.H != L.y.W || ls < 1) return hipErrorInvalidValue;
    a.gate_ls = ls;
  }
  a.N = L.x.N; a.S = L.x.H; a.Z = L.x.Z; a.Cbi = L.x.Cb; a.flags = L.flags;
  a.y_h = L.y_h; a.yh_nstride = L.yh_nstride;
  a.res_h = L.res_h ? L.res_h->p : nullptr; a.res_h_nstride = L.res_h ? L.res_h->nstride : 0;
  ah.fuse = 0; ah.bid0 = 0;
  ah.x_nstride_e = L.x.nstride; ah.x_plane_e = (long)L.x.Z * L.x.H * L.x.W * 8; ah.Cbp = L.x.Cb / 2;
  ConcatX cx;
  cx.nsrc = L.nsrc; cx.p1 = L.p1; cx.p2 = L.p2; cx.cbtot = 0;
  const uint16_t* xp_[3] = {nullptr, nullptr, nullptr};
  long xns_[3] = {0, 0, 0};
  int xcb_[3] = {0, 0x7fffffff, 0x7fffffff}, xcol_[3] = {0, 0, 0};
  if (L.nsrc) {
    // concat input: L.x carries the geometry (N, Z, H, W) and the PADDED (even) block count of the virtual tensor
    if (L.nsrc < 1 || L.nsrc > 3) return hipErrorInvalidValue;
    int cb = 0;
    for (int i = 0; i < L.nsrc; ++i) {
      if (!L.xs[i].p || L.xs[i].Cb < 1) return hipErrorInvalidValue;
      if (L.xs_collage[i] && (L.p1 < 2 || L.p2 < 2 || L.x.N % ((L.p1 - 1) * (L.p2 - 1)))) return hipErrorInvalidValue;
      xp_[i] = L.xs[i].p; xns_[i] = L.xs[i].nstride; xcb_[i] = cb; xcol_[i] = L.xs_collage[i];
      cb += L.xs[i].Cb;
    }
    cx.cbtot = cb;
    if (L.x.Cb != (cb + 1) / 2 * 2) return hipErrorInvalidValue;
    if (cx.p1 < 2) { cx.p1 = 2; cx.p2 = 2; }              // unused without a collaged source; keeps the index math finite
  }
  cx.p0 = xp_[0]; cx.p1p = xp_[1]; cx.p2p = xp_[2]; cx.ns0 = xns_[0]; cx.ns1 = xns_[1]; cx.ns2 = xns_[2];
  cx.cb1 = xcb_[1]; cx.cb2 = xcb_[2]; cx.col0 = xcol_[0]; cx.col1 = xcol_[1]; cx.col2 = xcol_[2];
  if ((L.x.Cb & 1) || L.x.H != L.x.W || L.y.H != L.x.H || L.y.Z != L.x.Z || L.y.N != L.x.N || (L.flags & EPI_UP2))
    return hipErrorInvalidValue;
  const int TN = conv_bf16_tn(L.Cout);
  a.ntile = (L.Cout + TN - 1) / TN;
  if (L.y.Cb > a.ntile * (TN / 8)) return hipErrorInvalidValue;
  const long vox = (long)a.N * a.Z * a.S * a.S;
#define TM_LAUNCH1H(TN_, NWV_)                                                                   \
  do {                                                                                          \
    using G = H1Geo<TN_, NWV_>;                                                                 \
    static DevOnce attr_done;                                                              \
    if (attr_done.need()) {                                                                           \
      const void* fns[6] = {(const void*)conv1_bf16<TN_, NWV_, false, 0>, (const void*)conv1_bf16<TN_, NWV_, false, 1>, \
                            (const void*)conv1_bf16<TN_, NWV_, false, 2>, (const void*)conv1_bf16<TN_, NWV_, false, 3>, \
                            (const void*)conv1_bf16<TN_, NWV_, true, 0>, (const void*)conv1_bf16<TN_, NWV_, true, 1>};  \
      for (const void* fn : fns) {                                                              \
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES); \
        if (e != hipSuccess) return e;                                                          \
      }                                                                                         \
      attr_done.mark();                                                                         \
    }                                                                                           \
    const long grid = ((vox + G::TM - 1) / G::TM) * a.ntile;                                    \
    const dim3 g_((unsigned)grid), b_(G::NT);                                                   \
    if (cx.nsrc) {                                                                              \
      if (ek == 1) hipLaunchKernelGGL((conv1_bf16<TN_, NWV_, true, 1>), g_, b_, G::LDS_BYTES, s, ah, zp, cx); \
      else hipLaunchKernelGGL((conv1_bf16<TN_, NWV_, true, 0>), g_, b_, G::LDS_BYTES, s, ah, zp, cx); \
    } else if (ek == 1) hipLaunchKernelGGL((conv1_bf16<TN_, NWV_, false, 1>), g_, b_, G::LDS_BYTES, s, ah, zp, cx); \
    else if (ek == 2) hipLaunchKernelGGL((conv1_bf16<TN_, NWV_, false, 2>), g_, b_, G::LDS_BYTES, s, ah, zp, cx); \
    else if (ek == 3) hipLaunchKernelGGL((conv1_bf16<TN_, NWV_, false, 3>), g_, b_, G::LDS_BYTES, s, ah, zp, cx); \
    else hipLaunchKernelGGL((conv1_bf16<TN_, NWV_, false, 0>), g_, b_, G::LDS_BYTES, s, ah, zp, cx); \
  } while (0)
  // epilogue kind (conv1_epilogue_stream / _gated): 16-bit stream output without fp32 side inputs; TM_CONV1_EK=0 forces the
  // generic one (A/B)
  static const int env_ek = [] { const char* e = getenv("TM_CONV1_EK"); return e ? atoi(e) : 2; }();   // 1: kinds 1 and 2 only
  const bool stream = env_ek != 0 && a.y_h && !a.gate && !a.res;
  const bool gelu = (a.flags & EPI_GELU) != 0;
  int ek = 0;
  if (stream && a.gate_h && a.res_h && !gelu && !cx.nsrc && env_ek == 2) ek = 3;
  else if (stream && !a.gate_h && !a.res_h && gelu && !cx.nsrc) ek = 2;
  else if (stream && !a.gate_h && !a.res_h && !gelu) ek = 1;
  static const int env1 = env_waves("TM_CONV1_WAVES");
  const int fw1 = L.force_waves ? L.force_waves : env1;
  if (fw1 != 0 && fw1 != 4 && fw1 != 8) return hipErrorInvalidValue;
  const bool small_wg = fw1 != 8;                       // default: two co-resident 4-wave workgroups per CU
  if (TN == 64) { if (small_wg) TM_LAUNCH1H(64, 4); else TM_LAUNCH1H(64, 8); }
  else { if (small_wg) TM_LAUNCH1H(128, 4); else TM_LAUNCH1H(128, 8); }
#undef TM_LAUNCH1H
  return hipGetLastError();
}

// compute units of the current device (workgroups of the 8-wave conv form that run at once), cached per device
static long cu_count() {
  static std::atomic<int> cus[128];
  const int d = DevOnce::dev();
  if (d < 0 || d >= 128) return 256;
  int c = cus[d].load();
  if (!c) {
    if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || c < 1) c = 256;
    cus[d].store(c);
  }
  return c;
}

hipError_t launch_conv27_bf16(const ConvLaunchH& L, hipStream_t s) {
  ConvArgsH ah;
  ConvArgs& a = ah.c;
  a.x = (const float*)L.x.p; a.x_nstride = 0; a.x_plane = 0;
  a.w = (const float*)L.w; a.bias = L.bias;
  a.y = L.y.p; a.y_nstride = L.y.nstride; a.y_plane = L.y.plane(); a.Cob = L.y.Cb;
  a.res = L.res ? L.res->p : nullptr; a.res_nstride = L.res ? L.res->nstride : 0;
  a.gate = nullptr; a.gate_nstride = 0;
  a.N = L.x.N; a.S = L.x.H; a.Z = L.x.Z; a.Cbi = L.x.Cb; a.flags = 0;
  a.y_h = L.y_h; a.yh_nstride = L.yh_nstride;
  a.res_h = L.res_h ? L.res_h->p : nullptr; a.res_h_nstride = L.res_h ? L.res_h->nstride : 0;
  ah.bid0 = 0;
  ah.fuse = L.fuse_norm; ah.norm_w = L.norm_w; ah.mod_scale = L.mod_scale; ah.mod_shift = L.mod_shift;
  ah.mod_stride = L.mod_stride; ah.per_image = L.per_image; ah.inv_c = 1.0f / (float)L.Cout;
  ah.a2 = L.a2.p; ah.a2_nstride = L.a2.nstride;
  ah.x_nstride_e = L.x.nstride; ah.x_plane_e = (long)L.x.Z * L.x.H * L.x.W * 8; ah.Cbp = L.x.Cb / 2;
  if (L.x.Cb & 1 || L.x.Z < 1 || L.y.Z != L.x.Z || L.x.H != L.x.W || L.y.H != (L.ups ? 2 : 1) * L.x.H || L.y.N != L.x.N)
    return hipErrorInvalidValue;
  const int S = a.S, TN = conv_bf16_tn(L.Cout);
  a.ntile = (L.Cout + TN - 1) / TN;
  if (L.res_half) {
    int ls = 0;
    while ((1 << ls) < L.y.H) ++ls;
    if (!L.res_h || (1 << ls) != L.y.H || L.y.H != L.y.W || ls < 1) return hipErrorInvalidValue;
    a.res_ls = ls;
  }
  // TM_CONV27_PP=0 (environment, read once) or force_waves == 9: the lockstep 8-wave kernel instead of the ping-pong one (A/B)
  static const int env_pp = [] { const char* e = getenv("TM_CONV27_PP"); return e ? atoi(e) : 1; }();
  const bool use_pp = env_pp != 0 && L.force_waves != 9;
  if (L.ups) {
    // upsampled-input form: TN = 128 only (the up blocks of the model family have Cout >= 128), Z == 2, no residual
    if (TN != 128 || L.x.Z != 2 || L.res || L.res_h || (S != 8 && S != 16 && S != 32 && S != 64)) return hipErrorInvalidValue;
    if (L.y.Cb > a.ntile * (TN / 8)) return hipErrorInvalidValue;
    if (L.fuse_norm && (a.ntile != 1 || L.Cout != TN || L.a2.Cb != TN / 8)) return hipErrorInvalidValue;
#define TM_LAUNCHU(TW_, NWV_)                                                                    \
  do {                                                                                          \
    using G = HGeo<128, TW_, NWV_, true>;                                                       \
    static_assert(G::LDS_BYTES <= 160 * 1024, "LDS budget of the upsampled-input form");        \
    static DevOnce attr_done;                                                                   \
    if (attr_done.need()) {                                                                     \
      hipError_t e = hipFuncSetAttribute((const void*)conv27_bf16<128, TW_, false, NWV_, true>, \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES); \
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv27_bf16<128, TW_, true, NWV_, true>, \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES); \
      if (e != hipSuccess) return e;                                                            \
      attr_done.mark();                                                                         \
    }                                                                                           \
    const long tiles = (long)(S / TW_) * (S / G::TR);                                           \
    const long pgs = (a.N + G::NPB - 1) / G::NPB;                                               \
    const long grid = pgs * a.Z * tiles * 4 * a.ntile;                                          \
    if (ah.fuse) hipLaunchKernelGGL((conv27_bf16<128, TW_, true, NWV_, true>), dim3((unsigned)grid), dim3(G::NT), G::LDS_BYTES, s, ah); \
    else hipLaunchKernelGGL((conv27_bf16<128, TW_, false, NWV_, true>), dim3((unsigned)grid), dim3(G::NT), G::LDS_BYTES, s, ah); \
  } while (0)
#define TM_LAUNCHUPP(TW_)                                                                        \
  do {                                                                                          \
    using G = HGeo<128, TW_, 8, true>;                                                          \
    static DevOnce attr_done;                                                                   \
    if (attr_done.need()) {                                                                     \
      hipError_t e = hipFuncSetAttribute((const void*)conv27_pp<128, TW_, false, true>,         \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES); \
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv27_pp<128, TW_, true, true>, \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES); \
      if (e != hipSuccess) return e;                                                            \
      attr_done.mark();                                                                         \
    }                                                                                           \
    const long tiles = (long)(S / TW_) * (S / G::TR);                                           \
    const long pgs = (a.N + G::NPB - 1) / G::NPB;                                               \
    const long grid = pgs * a.Z * tiles * 4 * a.ntile;                                          \
    if (ah.fuse) hipLaunchKernelGGL((conv27_pp<128, TW_, true, true>), dim3((unsigned)grid), dim3(G::NT), G::LDS_BYTES, s, ah); \
    else hipLaunchKernelGGL((conv27_pp<128, TW_, false, true>), dim3((unsigned)grid), dim3(G::NT), G::LDS_BYTES, s, ah); \
  } while (0)
    if (S >= 32) { if (use_pp) TM_LAUNCHUPP(32); else TM_LAUNCHU(32, 8); }
    else if (S == 16) { if (use_pp) TM_LAUNCHUPP(16); else TM_LAUNCHU(16, 8); }
    else TM_LAUNCHU(8, 4);
#undef TM_LAUNCHUPP
#undef TM_LAUNCHU
    return hipGetLastError();
  }
  if (L.y.Cb > a.ntile * (TN / 8)) return hipErrorInvalidValue;
  if (L.fuse_norm && (a.ntile != 1 || L.Cout != TN || L.a2.Cb != TN / 8 || L.res || L.res_h)) return hipErrorInvalidValue;
  if (S != 8 && S != 16 && S != 32 && S != 64 && S != 128) return hipErrorInvalidValue;
  static const int env27 = env_waves("TM_CONV27_WAVES");
  static const int tail_split = [] { const char* e = getenv("TM_CONV27_TAIL_SPLIT"); return e ? atoi(e) : 80; }();     // A/B switch (0 off, 1 <= ncu, else percent of 2 ncu)
  const long ncu = cu_count();
  long grid_override = 0;
  // TM_CONV27_K32=0 (environment, read once): the 32x32x16 ping-pong kernel instead of the 16x16x32 one (A/B); Z != 2 always
  static const int env_k32 = [] { const char* e = getenv("TM_CONV27_K32"); return e ? atoi(e) : 1; }();
  const bool use_k32 = env_k32 != 0 && a.Z == 2;
  const int fw27 = L.force_waves == 9 ? 8 : (L.force_waves ? L.force_waves : env27);
  if (fw27 != 0 && fw27 != 4 && fw27 != 8) return hipErrorInvalidValue;
#define TM_LAUNCHH(TN_, TW_)                                                                     \
  do {                                                                                          \
    using G8 = HGeo<TN_, TW_, 8>;                                                               \
    const long tiles8 = (long)(S / TW_) * (S / G8::TR);                                         \
    const long grid8 = ((a.N + G8::NPB - 1) / G8::NPB) * a.Z * tiles8 * a.ntile;                \
    const bool w8 = fw27 ? fw27 == 8 : grid8 >= 256;                                            \
    /* TAIL SPLIT (automatic form only).  One 8-wave workgroup fills a CU, so a launch runs in rounds of `ncu` workgroups */ \
    /* and its last, partial round costs a whole round: 632 tiles on 256 CUs are 3 rounds for 2.47 rounds of work.  When  */ \
    /* that round would occupy at most half the CUs, its tiles go to a second launch of the 4-wave form instead (half the */ \
    /* voxels per workgroup, twice the workgroups, still one per CU): the same voxels, outputs and arithmetic per output,   */ \
    /* finished in about half a round.  The split sits on a patch-group boundary so both forms tile the same voxel set.    */ \
    using G4 = HGeo<TN_, TW_, 4>;                                                               \
    const long unit = (long)a.ntile * a.Z * tiles8;                                             \
    const long full = grid8 / ncu * ncu / unit * unit;                                          \
    const long grid4 = ((a.N + G4::NPB - 1) / G4::NPB) * a.Z * ((long)(S / TW_) * (S / G4::TR)) * a.ntile; \
    const long tail4 = grid4 - 2 * full;                                                        \
    /* Two 4-wave workgroups fit a CU side by side (2 x LDS_BYTES <= 160 KB), so a tail of up to tail_pct % of 2 ncu half-tiles */ \
    /* is still resident all at once (TM_CONV27_TAIL_SPLIT = that percentage; 1 = the round-3a rule: at most ncu).            */ \
    const long tail_cap = tail_split == 1 ? ncu : (2 * G4::LDS_BYTES <= 160 * 1024 ? 2 * ncu * tail_split / 100 : ncu);  \
    if (w8 && use_pp && fw27 == 0 && tail_split && full >= ncu && tail4 > 0 && tail4 <= tail_cap) {  \
      grid_override = full;                                                                     \
      TM_LAUNCHPP(TN_, TW_);                                                                    \
      ah.bid0 = (int)(2 * full); grid_override = tail4;                                         \
      TM_LAUNCHHW(TN_, TW_, 4);                                                                 \
      ah.bid0 = 0; grid_override = 0;                                                           \
    } else if (w8 && use_pp) TM_LAUNCHPP(TN_, TW_); else if (w8) TM_LAUNCHHW(TN_, TW_, 8); else TM_LAUNCHHW(TN_, TW_, 4); \
  } while (0)
#define TM_LAUNCHPP(TN_, TW_)                                                                    \
  do {                                                                                          \
    using G = HGeo<TN_, TW_, 8>;                                                                \
    static DevOnce attr_done;                                                                   \
    if (attr_done.need()) {                                                                     \
      hipError_t e = hipFuncSetAttribute((const void*)conv27_pp<TN_, TW_, false>,               \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES); \
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv27_pp<TN_, TW_, true>,      \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES); \
      if (e != hipSuccess) return e;                                                            \
      attr_done.mark();                                                                         \
    }                                                                                           \
    const long tiles = (long)(S / TW_) * (S / G::TR);                                           \
    const long pgs = (a.N + G::NPB - 1) / G::NPB;                                               \
    const long grid = grid_override ? grid_override : pgs * a.Z * tiles * a.ntile;              \
    if (use_k32) {                                                                              \
      static DevOnce attr16_done;                                                               \
      if (attr16_done.need()) {                                                                 \
        hipError_t e = hipFuncSetAttribute((const void*)conv27_pp16<TN_, TW_, false>,           \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES); \
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv27_pp16<TN_, TW_, true>,  \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES); \
        if (e != hipSuccess) return e;                                                          \
        attr16_done.mark();                                                                     \
      }                                                                                         \
      if (ah.fuse) hipLaunchKernelGGL((conv27_pp16<TN_, TW_, true>), dim3((unsigned)grid), dim3(G::NT), G::LDS_BYTES, s, ah); \
      else hipLaunchKernelGGL((conv27_pp16<TN_, TW_, false>), dim3((unsigned)grid), dim3(G::NT), G::LDS_BYTES, s, ah); \
    } else if (ah.fuse) hipLaunchKernelGGL((conv27_pp<TN_, TW_, true>), dim3((unsigned)grid), dim3(G::NT), G::LDS_BYTES, s, ah); \
    else hipLaunchKernelGGL((conv27_pp<TN_, TW_, false>), dim3((unsigned)grid), dim3(G::NT), G::LDS_BYTES, s, ah); \
  } while (0)
#define TM_LAUNCHHW(TN_, TW_, NWV_)                                                              \
  do {                                                                                          \
    using G = HGeo<TN_, TW_, NWV_>;                                                             \
    static DevOnce attr_done;                                                              \
    if (attr_done.need()) {                                                                           \
      hipError_t e = hipFuncSetAttribute((const void*)conv27_bf16<TN_, TW_, false, NWV_>,       \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES); \
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv27_bf16<TN_, TW_, true, NWV_>, \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES); \
      if (e != hipSuccess) return e;                                                            \
      attr_done.mark();                                                                         \
    }                                                                                           \
    const long tiles = (long)(S / TW_) * (S / G::TR);                                           \
    const long pgs = (a.N + G::NPB - 1) / G::NPB;                                               \
    const long grid = grid_override ? grid_override : pgs * a.Z * tiles * a.ntile;              \
    if (ah.fuse) hipLaunchKernelGGL((conv27_bf16<TN_, TW_, true, NWV_>), dim3((unsigned)grid), dim3(G::NT), G::LDS_BYTES, s, ah); \
    else hipLaunchKernelGGL((conv27_bf16<TN_, TW_, false, NWV_>), dim3((unsigned)grid), dim3(G::NT), G::LDS_BYTES, s, ah); \
  } while (0)
  if (TN == 64) {
    if (S >= 32) TM_LAUNCHH(64, 32); else if (S == 16) TM_LAUNCHH(64, 16); else TM_LAUNCHH(64, 8);
  } else {
    if (S >= 32) TM_LAUNCHH(128, 32); else if (S == 16) TM_LAUNCHH(128, 16); else TM_LAUNCHH(128, 8);
  }
#undef TM_LAUNCHPP
#undef TM_LAUNCHHW
#undef TM_LAUNCHH
  return hipGetLastError();
}

}  // namespace tmk
