// Hand-written gfx950 (CDNA4) kernels of the Tera-MIND denoising hot path.
//
// Layout "CB8": fp32 [N][Cb][Z][H][W][8] -- channel blocks of 8 so that (a) an MFMA operand
// fragment (8 channels of one voxel) is one 32-byte piece, (b) a row of voxels is contiguous
// for coalesced HBM traffic, (c) a channel concat is a list of block ranges.
//
// Kernels (reference op each one replaces is cited at its definition):
//   conv3d_mfma / conv1_mfma   implicit-GEMM Conv3d on v_mfma_f32_32x32x2_f32 (exact fp32)
//   prep_kernel                concat + collage/up/down gather + RMSNorm(C) + modulate + SiLU
//   conv_direct_kernel         small convs (stem, head, RNA path) on VALU
//   (gene-gene and windowed attention kernels: tm_attn.hip)
//   time_embed / emb_all       timestep embedding MLP and all ResBlock emb_layers at once
//   (sampler_step / pad_patchify live in tm_sampler.hip: built with -ffp-contract=off)
#include "tm_device.h"

#include <stdlib.h>
#include <string.h>

namespace tmk {

// NZI selects the z structure of the k x 3 x 3 kernel:
//   NZI = 2  3x3x3, pad (1,1,1), Z == 2: the z-skip form above (18 of 27 taps per output plane)
//   NZI = 1  1x3x3, pad (0,1,1): in-plane conv, any Z   (RNA pyramid, model/unet_ours.py:290-295)
//   NZI = 3  3x3x3 over three staged planes zo + zi + zoff: zoff = 0 is valid-in-z (Zin = Zout + 2: down_z,
//            model/MBAblocks.py:472-474), zoff = -1 with planes outside [0, Zin) zero is pad (1,1,1) for any Z
//            (the ResBlock convs of the z_size 4 / 8 configs, rna_slc 8 / 16)
//   UPS (with NZI = 2)  the same 3x3x3 conv applied to a nearest-x2 UPSAMPLED input (ResBlock(up=True): Upsample then
//            in_layers, model/MBAblocks.py:254-258, blocks.py:362-371), computed on the LOW-resolution tensor: output voxel
//            (2y + py, 2x + px) sees only the 2 x 2 low-resolution voxels (y + py - 1 .. y + py, x + px - 1 .. x + px), so
//            each of the four phases (py, px) is a conv with 2 x 2 in-plane taps whose weights are the sums of the 3 x 3
//            taps that land on the same source voxel (packed per phase by conv_pack_ups_host): 8 instead of 18 taps per
//            output plane and a quarter of the input bytes; the zero padding of the upsampled tensor maps onto the zero
//            halo of the low-resolution tile.  Same sums, different association (weights are added before the products).
template <int NZI, int WM, int TW, bool UPS = false>
struct C3Geo {
  static constexpr int MV = 4 * WM * 32;                         // voxels per workgroup
  static constexpr int TR = (MV / TW < TW) ? (MV / TW) : TW;     // tile rows (<= plane size S == TW or 2*TW)
  static constexpr int NPB = MV / (TR * TW);                     // patches per workgroup
  static constexpr int HR = TR + 2, HC = TW + 2;
  static constexpr int XV = NPB * NZI * HR * HC;                 // halo voxels
  static constexpr int XPIECES = XV * 2;                         // 16-byte pieces
  static constexpr int PX = (XPIECES + 255) / 256;
  static constexpr int TZ = UPS ? 4 : 9;                         // in-plane taps
  static constexpr int NTAP = TZ * NZI;                          // taps staged per channel block
  static constexpr int WFLOATS = NTAP * 512;
  static constexpr int WPIECES = WFLOATS / 4;
  static constexpr int PW = (WPIECES + 255) / 256;
  static constexpr int TAPS_TOTAL = (NZI == 1) ? 9 : 3 * TZ;     // taps per channel block in the packed weights
  // channel blocks per stage: the 9-tap in-plane form (and the 8-tap upsampled form) has half the MFMAs per block of the
  // 18-tap z-skip form, so it stages two blocks per barrier pair to keep the same MFMA phase length
  static constexpr int KB = (NZI == 1 || UPS) ? 2 : 1;
  static constexpr int LDS_BYTES = KB * (WFLOATS + XV * 8) * 4;
};

template <int NZI, int WM, int TW, bool UPS = false>
__global__ __launch_bounds__(256, 2) void conv3d_mfma(ConvArgs a) {
  using G = C3Geo<NZI, WM, TW, UPS>;
  static_assert(!UPS || NZI == 2, "the upsampled-input form exists for the z-skip conv only");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int KB = G::KB;
  float* lw = lds;                                               // [KB][NTAP][512]
  float* lx = lds + KB * G::WFLOATS;                             // [KB][XV][8]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int i32 = lane & 31, h = lane >> 5;

  const int S = a.S;
  const int tiles_c = S / TW, tiles_r = S / G::TR;
  const int tiles = tiles_c * tiles_r;
  const int bid = xcd_swizzle(blockIdx.x, gridDim.x);
  const int nt = bid % a.ntile;
  int mt_ = bid / a.ntile;
  int py = 0, px = 0;                                            // UPS: output phase of this workgroup (the four phases of a
  if (UPS) { py = (mt_ >> 1) & 1; px = mt_ & 1; mt_ >>= 2; }     // tile are neighbours in the grid: they share the input halo)
  const int pg = mt_ / (a.Z * tiles);                            // a.Z = OUTPUT planes
  mt_ -= pg * a.Z * tiles;
  const int zo = mt_ / tiles;
  mt_ -= zo * tiles;
  const int tr = mt_ / tiles_c, tc = mt_ - tr * tiles_c;

  // ---- per-thread staging descriptors (constant over the K loop) ----
  long xoff[G::PX];
#pragma unroll
  for (int k = 0; k < G::PX; ++k) {
    const int i = tid + k * 256;
    long off = -1;
    if (i < G::XPIECES) {
      const int half = i & 1;
      int v = i >> 1;
      const int hc = v % G::HC; v /= G::HC;
      const int hr = v % G::HR; v /= G::HR;
      const int zi = v % NZI;
      const int ps = v / NZI;
      const int zs = (NZI == 2) ? zi : zo + zi + a.zoff;         // source plane
      const int n = pg * G::NPB + ps;
      const int y = tr * G::TR + hr - 1, x = tc * TW + hc - 1;
      if (n < a.N && y >= 0 && y < S && x >= 0 && x < S && (NZI == 2 || (zs >= 0 && zs < a.Zin)))
        off = (long)n * a.x_nstride + ((long)(zs * S + y) * S + x) * 8 + half * 4;
    }
    xoff[k] = off;
  }
  // weights: the staged taps of this (n-tile, cblk) are contiguous; NZI == 2 starts at kz = 1 - zo
  const int tap0 = (NZI == 2) ? (1 - zo) * G::TZ : 0;
  const float* wsrc = a.w + (((long)(UPS ? (py * 2 + px) * a.ntile : 0) + nt) * a.Cbi * G::TAPS_TOTAL + tap0) * 512 + tid * 4;
  const long w_cb_stride = G::TAPS_TOTAL * 512;

  // ---- per-lane fragment addresses ----
  int xb[WM];
  int on[WM], ooff[WM];
#pragma unroll
  for (int mt = 0; mt < WM; ++mt) {
    const int v = (wv * WM + mt) * 32 + i32;
    const int ps = v / (G::TR * TW);
    const int rem = v - ps * (G::TR * TW);
    const int r = rem / TW, c = rem - r * TW;
    xb[mt] = ((ps * NZI * G::HR + r) * G::HC + c) * 8 + 4 * h;
    const int n = pg * G::NPB + ps;
    on[mt] = n;
    const int y = tr * G::TR + r, x = tc * TW + c;
    if (n < a.N) {
      if (UPS) ooff[mt] = ((zo * 2 * S + 2 * y + py) * 2 * S + 2 * x + px) * 8;
      else if (a.flags & EPI_UP2) ooff[mt] = ((zo * 2 * S + 2 * y) * 2 * S + 2 * x) * 8;
      else ooff[mt] = ((zo * S + y) * S + x) * 8;
    } else ooff[mt] = -1;
  }
  const int wb = i32 * 8 + 4 * h;

  f32x16 acc[2][WM];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int mt = 0; mt < WM; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ct][mt][r] = 0.f;

  f32x4 xr[KB][G::PX], wr[KB][G::PW];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  auto load_stage = [&](int st) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const int cb = st * KB + kb;
      const bool live = KB == 1 || cb < a.Cbi;                   // a missing second block contributes zeros
      const float* xp = a.x + (long)cb * a.x_plane;
#pragma unroll
      for (int k = 0; k < G::PX; ++k) xr[kb][k] = (live && xoff[k] >= 0) ? *(const f32x4*)(xp + xoff[k]) : zero4;
      const float* wp = wsrc + (long)cb * w_cb_stride;
#pragma unroll
      for (int k = 0; k < G::PW; ++k)
        if (G::WPIECES % 256 == 0 || tid + k * 256 < G::WPIECES) wr[kb][k] = live ? *(const f32x4*)(wp + k * 1024) : zero4;
    }
  };

  const int nst = (a.Cbi + KB - 1) / KB;
  load_stage(0);
  for (int st = 0; st < nst; ++st) {
    __syncthreads();
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
      for (int k = 0; k < G::PX; ++k)
        if (tid + k * 256 < G::XPIECES) *(f32x4*)(lx + kb * G::XV * 8 + (tid + k * 256) * 4) = xr[kb][k];
#pragma unroll
      for (int k = 0; k < G::PW; ++k)
        if (G::WPIECES % 256 == 0 || tid + k * 256 < G::WPIECES) *(f32x4*)(lw + kb * G::WFLOATS + (tid + k * 256) * 4) = wr[kb][k];
    }
    __syncthreads();
    if (st + 1 < nst) load_stage(st + 1);

    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
      for (int zi = 0; zi < NZI; ++zi) {
#pragma unroll
        for (int ky = 0; ky < (UPS ? 2 : 3); ++ky) {
#pragma unroll
          for (int kx = 0; kx < (UPS ? 2 : 3); ++kx) {
            const int tap = UPS ? zi * 4 + ky * 2 + kx : zi * 9 + ky * 3 + kx;
            const int xd = kb * G::XV * 8 + ((zi * G::HR + ky + py) * G::HC + kx + px) * 8;
            f32x4 wf[2], xf[WM];
            wf[0] = *(const f32x4*)(lw + kb * G::WFLOATS + tap * 512 + wb);
            wf[1] = *(const f32x4*)(lw + kb * G::WFLOATS + tap * 512 + 256 + wb);
#pragma unroll
            for (int mt = 0; mt < WM; ++mt) xf[mt] = *(const f32x4*)(lx + xb[mt] + xd);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
              for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int mt = 0; mt < WM; ++mt)
                  acc[ct][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[ct][kk], xf[mt][kk], acc[ct][mt], 0, 0, 0);
          }
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
  }
  conv_epilogue<WM>(a, acc, nt, h, on, ooff, 2 * S);
}

// ---- 1x1x1 conv / Linear over voxels (flat voxel tiles, KC channel blocks per stage) ----
// Replaces the skip_connection Conv3d(k=1) (model/MBAblocks.py:220-224) and every nn.Linear
// of AttnBlock / Attention / Mlp applied to '(z h w) c' tokens (model/MBAblocks.py:465,
// 538-544; timm Mlp fc1/fc2).
// WN = cout tiles of 32 per wave: 2 (64-cout workgroup tile) or 4 (128-cout tile: half the activation bytes per
// FLOP -- at 64 couts this kernel is bound by the L2 -> LDS staging rate, not by the matrix pipe)
template <int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv1_mfma(ConvArgs a) {
  constexpr int MV = 4 * WM * 32;
  constexpr int KC = 4;
  constexpr int XP = KC * MV * 2 / 256;       // x pieces per thread
  constexpr int NT64 = WN / 2;                // 64-cout weight tiles per workgroup
  constexpr int WP = NT64 * KC * 64 * 2 / 256;
  __shared__ __attribute__((aligned(16))) float lds[NT64 * KC * 512 + KC * MV * 8];
  float* lw = lds;                            // [NT64][KC][64][8]
  float* lx = lds + NT64 * KC * 512;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int i32 = lane & 31, h = lane >> 5;
  const int bid = xcd_swizzle(blockIdx.x, gridDim.x);
  const int nt = bid % a.ntile;
  const int mtile = bid / a.ntile;
  const long VPN = (long)a.Z * a.S * a.S;      // voxels per n
  const long vtot = VPN * a.N;

  long xoff[XP];
  int xkc[XP];
#pragma unroll
  for (int k = 0; k < XP; ++k) {
    const int i = tid + k * 256;
    const int half = i & 1;
    const int v = (i >> 1) % MV;
    xkc[k] = (i >> 1) / MV;
    const long vg = (long)mtile * MV + v;
    long off = -1;
    if (vg < vtot) {
      const long n = vg / VPN;
      off = n * a.x_nstride + (vg - n * VPN) * 8 + half * 4;
    }
    xoff[k] = off;
  }
  // piece i = tid + k*256 of the weight stage: 64-cout tile i / (KC*128), then [KC][64][2] inside it
  const float* wsrc = a.w + (long)nt * NT64 * a.Cbi * 512;

  int xb[WM], on[WM], ooff[WM];
#pragma unroll
  for (int mt = 0; mt < WM; ++mt) {
    const int v = (wv * WM + mt) * 32 + i32;
    xb[mt] = v * 8 + 4 * h;
    const long vg = (long)mtile * MV + v;
    if (vg < vtot) {
      const long n = vg / VPN;
      on[mt] = (int)n;
      ooff[mt] = (int)((vg - n * VPN) * 8);
    } else { on[mt] = 0; ooff[mt] = -1; }
  }
  const int wb = i32 * 8 + 4 * h;

  f32x16 acc[WN][WM];
#pragma unroll
  for (int ct = 0; ct < WN; ++ct)
#pragma unroll
    for (int mt = 0; mt < WM; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ct][mt][r] = 0.f;

  f32x4 xr[XP], wr[WP];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  auto load_stage = [&](int cb0) {
#pragma unroll
    for (int k = 0; k < XP; ++k) {
      const int cb = cb0 + xkc[k];
      xr[k] = (xoff[k] >= 0 && cb < a.Cbi) ? *(const f32x4*)(a.x + (long)cb * a.x_plane + xoff[k]) : zero4;
    }
#pragma unroll
    for (int k = 0; k < WP; ++k) {
      const int i = tid + k * 256;
      const int t64 = i / (KC * 128), j = i - t64 * (KC * 128);
      const int cb = cb0 + j / 128;
      wr[k] = (cb < a.Cbi) ? *(const f32x4*)(wsrc + ((long)t64 * a.Cbi + cb0) * 512 + j * 4) : zero4;
    }
  };

  load_stage(0);
  for (int cb0 = 0; cb0 < a.Cbi; cb0 += KC) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < XP; ++k) *(f32x4*)(lx + (tid + k * 256) * 4) = xr[k];
#pragma unroll
    for (int k = 0; k < WP; ++k) *(f32x4*)(lw + (tid + k * 256) * 4) = wr[k];
    __syncthreads();
    if (cb0 + KC < a.Cbi) load_stage(cb0 + KC);
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      f32x4 wf[WN], xf[WM];
#pragma unroll
      for (int ct = 0; ct < WN; ++ct) wf[ct] = *(const f32x4*)(lw + (ct >> 1) * (KC * 512) + kc * 512 + (ct & 1) * 256 + wb);
#pragma unroll
      for (int mt = 0; mt < WM; ++mt) xf[mt] = *(const f32x4*)(lx + kc * MV * 8 + xb[mt]);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int ct = 0; ct < WN; ++ct)
#pragma unroll
          for (int mt = 0; mt < WM; ++mt)
            acc[ct][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[ct][kk], xf[mt][kk], acc[ct][mt], 0, 0, 0);
    }
  }
  conv_epilogue<WM, WN>(a, acc, nt, h, on, ooff, 0);
}

size_t conv_pack_floats(int Cout, int Cbi, int taps) {
  const int ntile = (Cout + 63) / 64;
  return (size_t)ntile * Cbi * taps * 512;
}

void conv_pack_host(const float* w, int Cout, const int* seg_c, int nseg, int taps, float* out) {
  int Cin = 0, Cbi = 0;
  for (int s = 0; s < nseg; ++s) { Cin += seg_c[s]; Cbi += (seg_c[s] + 7) / 8; }
  const int ntile = (Cout + 63) / 64;
  memset(out, 0, conv_pack_floats(Cout, Cbi, taps) * sizeof(float));
  int ci0 = 0, cb0 = 0;
  for (int s = 0; s < nseg; ++s) {
    for (int c = 0; c < seg_c[s]; ++c) {
      const int ci = ci0 + c;
      const int cb = cb0 + c / 8, c8 = c % 8;
      for (int co = 0; co < Cout; ++co) {
        const int nt = co / 64, col = co % 64;
        const float* src = w + ((size_t)co * Cin + ci) * taps;
        float* dst = out + (((size_t)nt * Cbi + cb) * taps) * 512 + (size_t)col * 8 + c8;
        for (int t = 0; t < taps; ++t) dst[(size_t)t * 512] = src[t];
      }
    }
    ci0 += seg_c[s];
    cb0 += (seg_c[s] + 7) / 8;
  }
  (void)ntile;
}

// Phase weights of the upsampled-input conv (C3Geo, UPS): for output phase (py, px) the 3 x 3 in-plane taps collapse onto a
// 2 x 2 window of the low-resolution input -- rows: py = 0: {ky 0} | {ky 1, 2}; py = 1: {ky 0, 1} | {ky 2}; columns alike.
// out: [phase = 2 py + px][conv_pack_host layout with 12 taps (kz, ky', kx')]
size_t conv_pack_ups_floats(int Cout, int Cbi) { return 4 * conv_pack_floats(Cout, Cbi, 12); }
void conv_pack_ups_host(const float* w /*[Cout][Cin][27]*/, int Cout, const int* seg_c, int nseg, float* out) {
  int Cin = 0, Cbi = 0;
  for (int s = 0; s < nseg; ++s) { Cin += seg_c[s]; Cbi += (seg_c[s] + 7) / 8; }
  static const int G0[2][2][2] = {{{0, 0}, {1, 2}}, {{0, 1}, {2, 2}}};       // [phase bit][window pos] -> tap range [lo, hi]
  const size_t per = conv_pack_floats(Cout, Cbi, 12);
  float* weff = (float*)malloc((size_t)Cout * Cin * 12 * sizeof(float));
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px) {
      for (size_t i = 0; i < (size_t)Cout * Cin; ++i)
        for (int kz = 0; kz < 3; ++kz)
          for (int wy = 0; wy < 2; ++wy)
            for (int wx = 0; wx < 2; ++wx) {
              float acc = 0.f;
              for (int ky = G0[py][wy][0]; ky <= G0[py][wy][1]; ++ky)
                for (int kx = G0[px][wx][0]; kx <= G0[px][wx][1]; ++kx) acc += w[i * 27 + kz * 9 + ky * 3 + kx];
              weff[i * 12 + kz * 4 + wy * 2 + wx] = acc;
            }
      conv_pack_host(weff, Cout, seg_c, nseg, 12, out + (size_t)(py * 2 + px) * per);
    }
  free(weff);
}

void vec_pack_host(const float* v, const int* seg_c, int nseg, float* out) {
  int ci0 = 0, cb0 = 0;
  for (int s = 0; s < nseg; ++s) {
    const int nb = (seg_c[s] + 7) / 8;
    for (int c = 0; c < nb * 8; ++c) out[cb0 * 8 + c] = (c < seg_c[s]) ? v[ci0 + c] : 0.f;
    ci0 += seg_c[s];
    cb0 += nb;
  }
}

hipError_t launch_conv_mfma(const ConvLaunch& L, hipStream_t s) {
  ConvArgs a;
  a.x = L.x.p; a.x_nstride = L.x.nstride; a.x_plane = L.x.plane();
  a.w = L.w.w; a.bias = L.w.bias;
  a.y = L.y.p; a.y_nstride = L.y.nstride; a.y_plane = L.y.plane(); a.Cob = L.y.Cb;
  a.res = L.res ? L.res->p : nullptr; a.res_nstride = L.res ? L.res->nstride : 0;
  a.gate = L.gate ? L.gate->p : nullptr; a.gate_nstride = L.gate ? L.gate->nstride : 0;
  if (L.gate_half) {
    int ls = 0;
    while ((1 << ls) < L.y.H) ++ls;
    if (!L.gate || L.w.taps != 1 || (1 << ls) != L.y.H || L.y.H != L.y.W || ls < 1) return hipErrorInvalidValue;
    a.gate_ls = ls;
  }
  a.N = L.x.N; a.S = L.x.H; a.Z = L.x.Z; a.Cbi = L.w.Cbi; a.ntile = L.w.ntile; a.flags = L.flags;
  if (L.x.Cb != L.w.Cbi || L.x.H != L.x.W) return hipErrorInvalidValue;
  if (L.y.Cb > L.w.ntile * 8 || L.y.N != L.x.N) return hipErrorInvalidValue;
  const long vox = (long)a.N * a.Z * a.S * a.S;
  if (L.w.taps == 1) {
    if (L.flags & EPI_UP2) return hipErrorInvalidValue;
    if (L.y.H != L.x.H || L.y.Z != L.x.Z) return hipErrorInvalidValue;
    int variant = L.tile_variant ? L.tile_variant : ((vox / 256) * a.ntile >= 512 ? 2 : 1);
    if (variant == 2 && (a.ntile & 1) == 0 && (vox / 256) * (a.ntile / 2) >= 512) {
      const long mt = (vox + 255) / 256;                  // 128-cout x 256-voxel workgroups
      a.ntile /= 2;
      hipLaunchKernelGGL((conv1_mfma<2, 4>), dim3((unsigned)(mt * a.ntile)), dim3(256), 0, s, a);
    } else if (variant == 2) {
      const long mt = (vox + 255) / 256;
      hipLaunchKernelGGL((conv1_mfma<2, 2>), dim3((unsigned)(mt * a.ntile)), dim3(256), 0, s, a);
    } else {
      const long mt = (vox + 127) / 128;
      hipLaunchKernelGGL((conv1_mfma<1, 2>), dim3((unsigned)(mt * a.ntile)), dim3(256), 0, s, a);
    }
    return hipGetLastError();
  }
  // ---- k x 3 x 3 kernels ----
  const int S = a.S;
  int nzi;
  a.Zin = L.x.Z; a.zoff = 0;
  if (L.zmode == ZM_PAD1 && L.w.taps == 9) {   // 3x3x3 pad 1 on ONE plane: only the centre z slice meets data (packed as 9 taps)
    if (L.x.Z != 1 || L.y.Z != 1) return hipErrorInvalidValue;
    nzi = 1;
  } else if (L.zmode == ZM_PAD1) {             // 3x3x3 pad 1
    if (L.w.taps != 27 || L.y.Z != L.x.Z) return hipErrorInvalidValue;
    if (L.x.Z == 2) nzi = 2;                   // z-skip form
    else { nzi = 3; a.zoff = -1; }
  } else if (L.zmode == ZM_INPLANE) {  // 1x3x3
    if (L.w.taps != 9 || L.y.Z != L.x.Z) return hipErrorInvalidValue;
    nzi = 1;
  } else if (L.zmode == ZM_VALID) {    // 3x3x3 valid in z
    if (L.w.taps != 27 || L.y.Z != L.x.Z - 2) return hipErrorInvalidValue;
    nzi = 3;
  } else if (L.zmode == ZM_UPS) {      // 3x3x3 pad 1 of the nearest-x2 upsampled x, on the low-resolution x (phase weights)
    if (L.w.taps != 12 || L.x.Z != 2 || L.y.Z != 2 || (L.flags & EPI_UP2) || L.res || L.gate) return hipErrorInvalidValue;
    nzi = 2;
  } else return hipErrorInvalidValue;
  a.Z = L.y.Z;
  if (S != 4 && S != 8 && S != 16 && S != 32 && S != 64 && S != 128) return hipErrorInvalidValue;
  if (((L.flags & EPI_UP2) || L.zmode == ZM_UPS) ? (L.y.H != 2 * S) : (L.y.H != S)) return hipErrorInvalidValue;
  if (L.res_half) {
    int ls = 0;
    while ((1 << ls) < L.y.H) ++ls;
    if (!L.res || (1 << ls) != L.y.H || L.y.H != L.y.W || ls < 1 || (L.flags & EPI_UP2)) return hipErrorInvalidValue;
    a.res_ls = ls;
  }
  const long ovox = (long)a.N * a.Z * a.S * a.S;
  if (L.zmode == ZM_UPS) {
#define TM_LAUNCHU(WM, TW)                                                                       \
  do {                                                                                          \
    using G = C3Geo<2, WM, TW, true>;                                                           \
    static DevOnce attr_once;                                                                   \
    if (attr_once.need()) {                                                                     \
      hipError_t e = hipFuncSetAttribute((const void*)conv3d_mfma<2, WM, TW, true>,             \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES); \
      if (e != hipSuccess) return e;                                                            \
      attr_once.mark();                                                                         \
    }                                                                                           \
    const long tiles = (long)(S / TW) * (S / G::TR);                                            \
    const long pgs = (a.N + G::NPB - 1) / G::NPB;                                               \
    const long grid = pgs * a.Z * tiles * 4 * a.ntile;                                          \
    hipLaunchKernelGGL((conv3d_mfma<2, WM, TW, true>), dim3((unsigned)grid), dim3(256), G::LDS_BYTES, s, a); \
  } while (0)
    const int variant = L.tile_variant ? L.tile_variant : ((ovox / 256) * 4 * a.ntile >= 512 ? 2 : 1);
    if (S < 8) { if (S != 4) return hipErrorInvalidValue; TM_LAUNCHU(1, 4); }
    else if (variant == 2) { if (S >= 32) TM_LAUNCHU(2, 32); else if (S == 16) TM_LAUNCHU(2, 16); else TM_LAUNCHU(2, 8); }
    else { if (S >= 32) TM_LAUNCHU(1, 32); else if (S == 16) TM_LAUNCHU(1, 16); else TM_LAUNCHU(1, 8); }
#undef TM_LAUNCHU
    return hipGetLastError();
  }
  int variant = L.tile_variant ? L.tile_variant : ((ovox / 256) * a.ntile >= 512 ? 2 : 1);
#define TM_LAUNCH3(NZI, WM, TW)                                                                  \
  do {                                                                                          \
    using G = C3Geo<NZI, WM, TW>;                                                               \
    static DevOnce attr_once;                                                                   \
    if (attr_once.need()) {                                                                     \
      hipError_t e = hipFuncSetAttribute((const void*)conv3d_mfma<NZI, WM, TW>,                 \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES); \
      if (e != hipSuccess) return e;                                                            \
      attr_once.mark();                                                                         \
    }                                                                                           \
    const long tiles = (long)(S / TW) * (S / G::TR);                                            \
    const long pgs = (a.N + G::NPB - 1) / G::NPB;                                               \
    const long grid = pgs * a.Z * tiles * a.ntile;                                              \
    hipLaunchKernelGGL((conv3d_mfma<NZI, WM, TW>), dim3((unsigned)grid), dim3(256), G::LDS_BYTES, s, a); \
  } while (0)
  if (nzi == 2) {
    if (S < 8) { if (S != 4) return hipErrorInvalidValue; TM_LAUNCH3(2, 1, 4); }
    else if (variant == 2) {
      if (S >= 32) TM_LAUNCH3(2, 2, 32); else if (S == 16) TM_LAUNCH3(2, 2, 16); else TM_LAUNCH3(2, 2, 8);
    } else {
      if (S >= 32) TM_LAUNCH3(2, 1, 32); else if (S == 16) TM_LAUNCH3(2, 1, 16); else TM_LAUNCH3(2, 1, 8);
    }
  } else if (nzi == 1) {
    if (S == 4) TM_LAUNCH3(1, 1, 4);
    else if (variant == 2) {
      if (S >= 32) TM_LAUNCH3(1, 2, 32); else if (S == 16) TM_LAUNCH3(1, 2, 16); else TM_LAUNCH3(1, 2, 8);
    } else {                                   // small grids: 128-voxel workgroups fill more CUs
      if (S >= 32) TM_LAUNCH3(1, 1, 32); else if (S == 16) TM_LAUNCH3(1, 1, 16); else TM_LAUNCH3(1, 1, 8);
    }
  } else {
    if (S == 4) TM_LAUNCH3(3, 1, 4);
    else if (variant == 2) {
      if (S >= 32) TM_LAUNCH3(3, 2, 32); else if (S == 16) TM_LAUNCH3(3, 2, 16); else TM_LAUNCH3(3, 2, 8);
    } else {
      if (S >= 32) TM_LAUNCH3(3, 1, 32); else if (S == 16) TM_LAUNCH3(3, 1, 16); else TM_LAUNCH3(3, 1, 8);
    }
  }
#undef TM_LAUNCH3
  return hipGetLastError();
}

// ==========================================================================================
// prep: gather (concat / collage / nearest-up / avg-down) + LlamaRMSNorm over C + modulate +
// SiLU, written as the activated conv input.  Replaces th.cat + to_collage
// (model/unet_ours.py:325-341,384,418), LlamaRMSNorm(dim=1) (model/MBAblocks.py:21-43), the
// scale/shift of apply_conditions (:356-367), modulate (:608-614), nn.SiLU and
// Upsample / Downsample (model/blocks.py:362-371,389-403) as they occur in ResBlock._forward
// (:254-261) and AttnBlock._forward (:484-489).
// 64 voxels per workgroup (lane = voxel: 2 KB contiguous per wave load), the 4 waves split
// the channel blocks and combine their sums of squares through LDS.
// ==========================================================================================
struct PrepArgs { PrepLaunch L; };

// CACHED: the voxel's channel blocks owned by this wave (<= 8) stay in registers between the
// sum-of-squares pass and the apply pass, so every source byte is read from HBM once.
template <int NSUB, bool CACHED>
__global__ __launch_bounds__(256) void prep_kernel(PrepArgs pa) {
  const PrepLaunch& L = pa.L;
  __shared__ float red[4][NSUB][64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int S = L.S, Z = L.Z;
  const long vpn = (long)Z * S * S;
  const long vidx = (long)blockIdx.x * 64 + lane;
  const bool valid = vidx < vpn * L.N;
  int n = 0, z = 0, y = 0, x = 0;
  if (valid) {
    n = (int)(vidx / vpn);
    int rem = (int)(vidx - (long)n * vpn);
    z = rem / (S * S); rem -= z * S * S;
    y = rem / S; x = rem - y * S;
  }
  // source offsets
  long soff[3][NSUB];
  long splane[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (k >= L.nsrc) { splane[k] = 0; continue; }
    int Ss = S;
    if (L.resample == RS_UP2) Ss = S / 2;
    if (L.resample == RS_DOWN2 || L.resample == RS_PICK2) Ss = S * 2;
    splane[k] = (long)Z * Ss * Ss * 8;
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
      // the voxel's coordinates in the source plane, then (collage) half a source patch down / right into the neighbour
      int ns = n, ys = y, xs = x;
      if (L.resample == RS_UP2) { ys = y >> 1; xs = x >> 1; }
      if (L.resample == RS_DOWN2) { ys = 2 * y + (sub >> 1); xs = 2 * x + (sub & 1); }
      if (L.resample == RS_PICK2) { ys = 2 * y; xs = 2 * x; }
      if (L.src[k].collage) {
        const int q1 = L.p1 - 1, q2 = L.p2 - 1;
        const int bi = n / (q1 * q2);
        const int q = n - bi * q1 * q2;
        int i = q / q2, j = q - i * q2;
        ys += Ss / 2; if (ys >= Ss) { ys -= Ss; i += 1; }
        xs += Ss / 2; if (xs >= Ss) { xs -= Ss; j += 1; }
        ns = bi * L.p1 * L.p2 + i * L.p2 + j;
      }
      soff[k][sub] = (long)ns * L.src[k].nstride + ((long)(z * Ss + ys) * Ss + xs) * 8;
    }
  }
  // this wave owns the virtual (concatenated) channel blocks gb = wv, wv + 4, wv + 8, ...
  const int cb0 = L.src[0].Cb, cb01 = cb0 + ((L.nsrc > 1) ? L.src[1].Cb : 0);
  const int cbtot = cb01 + ((L.nsrc > 2) ? L.src[2].Cb : 0);
  // 8 channels of one voxel of the virtual (concatenated) block gb: fp32 sources (32 bytes) or, with src_h, 16-bit
  // sources (16 bytes; offsets and strides are in elements either way)
  auto load8 = [&](int gb, int sub, f32x4& a0, f32x4& a1) {
    const int k = (gb >= cb0) + (gb >= cb01);
    const int cb = gb - (k == 0 ? 0 : (k == 1 ? cb0 : cb01));
    const long off = soff[k][sub] + (long)cb * splane[k];
    if (L.src_h) {
      const uint16_t* p = (const uint16_t*)L.src[k].p + off;
      if (L.h_f16) {
        typedef _Float16 f16x8_s __attribute__((ext_vector_type(8)));
        const f16x8_s v = *(const f16x8_s*)p;
        a0 = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
        a1 = f32x4{(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
      } else {
        typedef __bf16 bf16x8_s __attribute__((ext_vector_type(8)));
        const bf16x8_s v = *(const bf16x8_s*)p;
        a0 = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
        a1 = f32x4{(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
      }
    } else {
      const float* p = L.src[k].p + off;
      a0 = *(const f32x4*)p; a1 = *(const f32x4*)(p + 4);
    }
  };
  constexpr int NC = CACHED ? 8 : 1;
  f32x4 c0[NC], c1[NC];            // CACHED only (NSUB == 1)

  float rstd[NSUB];
#pragma unroll
  for (int sub = 0; sub < NSUB; ++sub) rstd[sub] = 1.f;
  if (L.norm_w || CACHED) {
    float ssq[NSUB];
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) ssq[sub] = 0.f;
    if (valid) {
      if (CACHED) {
#pragma unroll
        for (int i = 0; i < NC; ++i) {
          const int gb = wv + 4 * i;
          if (gb < cbtot) {
            load8(gb, 0, c0[i], c1[i]);
            ssq[0] += c0[i][0] * c0[i][0] + c0[i][1] * c0[i][1] + c0[i][2] * c0[i][2] + c0[i][3] * c0[i][3] +
                      c1[i][0] * c1[i][0] + c1[i][1] * c1[i][1] + c1[i][2] * c1[i][2] + c1[i][3] * c1[i][3];
          }
        }
      } else {
        for (int gb = wv; gb < cbtot; gb += 4) {
#pragma unroll
          for (int sub = 0; sub < NSUB; ++sub) {
            f32x4 a0, a1;
            load8(gb, sub, a0, a1);
            ssq[sub] += a0[0] * a0[0] + a0[1] * a0[1] + a0[2] * a0[2] + a0[3] * a0[3] +
                        a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2] + a1[3] * a1[3];
          }
        }
      }
    }
    if (L.norm_w) {
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub) red[wv][sub][lane] = ssq[sub];
      __syncthreads();
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub) {
        const float t = red[0][sub][lane] + red[1][sub][lane] + red[2][sub][lane] + red[3][sub][lane];
        rstd[sub] = 1.0f / sqrtf(t * L.inv_c + TM_EPS);
      }
    }
  }
  if (!valid) return;
  const long oplane = vpn * 8;
  const long oin = ((long)(z * S + y) * S + x) * 8;
  const int img = n / L.per_image;
  auto emit = [&](int gb, int i) {
    float wn[8], sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { wn[j] = 1.f; sc[j] = 0.f; sh[j] = 0.f; }
    if (L.norm_w) {
#pragma unroll
      for (int j = 0; j < 8; ++j) wn[j] = L.norm_w[gb * 8 + j];
    }
    if (L.mod == MOD_IMAGE) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        sc[j] = L.mod_scale[(long)img * L.mod_stride + gb * 8 + j];
        sh[j] = L.mod_shift[(long)img * L.mod_stride + gb * 8 + j];
      }
    } else if (L.mod == MOD_VOXEL) {
      const long mo = !L.mod_half ? (long)n * L.mod_stride + (long)gb * oplane + oin
                                  : (long)n * L.mod_stride + (long)gb * (oplane >> 2) +
                                        ((long)(z * (S >> 1) + (y >> 1)) * (S >> 1) + (x >> 1)) * 8;
      if (L.mod_scale_h) {
        if (L.h_f16) {
          typedef _Float16 f16x8_m __attribute__((ext_vector_type(8)));
          const f16x8_m scb = *(const f16x8_m*)(L.mod_scale_h + mo), shb = *(const f16x8_m*)(L.mod_shift_h + mo);
#pragma unroll
          for (int j = 0; j < 8; ++j) { sc[j] = (float)scb[j]; sh[j] = (float)shb[j]; }
        } else {
          typedef __bf16 bf16x8_m __attribute__((ext_vector_type(8)));
          const bf16x8_m scb = *(const bf16x8_m*)(L.mod_scale_h + mo), shb = *(const bf16x8_m*)(L.mod_shift_h + mo);
#pragma unroll
          for (int j = 0; j < 8; ++j) { sc[j] = (float)scb[j]; sh[j] = (float)shb[j]; }
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) { sc[j] = L.mod_scale[mo + j]; sh[j] = L.mod_shift[mo + j]; }
      }
    }
    float o[8], r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { o[j] = 0.f; r[j] = 0.f; }
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
      f32x4 a0, a1;
      if (CACHED) { a0 = c0[i]; a1 = c1[i]; }
      else load8(gb, sub, a0, a1);
      const float xv[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = xv[j];
        r[j] += v;
        if (L.norm_w) v = wn[j] * (v * rstd[sub]);
        if (L.mod != MOD_NONE) v = v * (1.0f + sc[j]) + sh[j];
        // SiLU on the hardware exp2 / rcp (1 ulp each, ~3e-7 relative on the result) in fp32 too: libm expf + an IEEE division
        // are ~30 VALU instructions per element of an otherwise HBM-bound pass (configs[1] fp32: 55.22 -> 54.63 ms same box)
        if (L.act) v = silu_h16(v);
        if (L.drop_mask) v *= L.drop_mask[(long)n * L.drop_ns + (long)gb * oplane + oin + j] * L.drop_scale;
        o[j] += v;
      }
    }
    if (NSUB == 4) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { o[j] *= 0.25f; r[j] *= 0.25f; }
    }
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    const long eo = (long)gb * oplane + oin;
    if (L.out_h) {
      if (L.h_f16) {
        f16x8 ob;
#pragma unroll
        for (int j = 0; j < 8; ++j) ob[j] = (_Float16)o[j];
        *(f16x8*)(L.out_h + (long)n * L.out_h_nstride + eo) = ob;
      } else {
        bf16x8 ob;
#pragma unroll
        for (int j = 0; j < 8; ++j) ob[j] = (__bf16)o[j];
        *(bf16x8*)(L.out_h + (long)n * L.out_h_nstride + eo) = ob;
      }
    } else if (L.out) {
      float* op = L.out + (long)n * L.out_nstride + eo;
      *(f32x4*)op = f32x4{o[0], o[1], o[2], o[3]};
      *(f32x4*)(op + 4) = f32x4{o[4], o[5], o[6], o[7]};
    }
    if (L.raw) {
      float* rp = L.raw + (long)n * L.raw_nstride + eo;
      *(f32x4*)rp = f32x4{r[0], r[1], r[2], r[3]};
      *(f32x4*)(rp + 4) = f32x4{r[4], r[5], r[6], r[7]};
    }
    if (L.raw_h) {
      if (L.h_f16) {
        f16x8 rb;
#pragma unroll
        for (int j = 0; j < 8; ++j) rb[j] = (_Float16)r[j];
        *(f16x8*)(L.raw_h + (long)n * L.raw_h_nstride + eo) = rb;
      } else {
        bf16x8 rb;
#pragma unroll
        for (int j = 0; j < 8; ++j) rb[j] = (__bf16)r[j];
        *(bf16x8*)(L.raw_h + (long)n * L.raw_h_nstride + eo) = rb;
      }
    }
  };
  if (CACHED) {
#pragma unroll
    for (int i = 0; i < NC; ++i)
      if (wv + 4 * i < cbtot) emit(wv + 4 * i, i);
  } else {
    for (int gb = wv; gb < cbtot; gb += 4) emit(gb, 0);
  }
  if (L.pad_blocks && wv == 0) {
    for (int pb = 0; pb < L.pad_blocks; ++pb) {
      if (L.out_h) *(uint4*)(L.out_h + (long)n * L.out_h_nstride + (long)(cbtot + pb) * oplane + oin) = uint4{0u, 0u, 0u, 0u};
      if (L.raw_h) *(uint4*)(L.raw_h + (long)n * L.raw_h_nstride + (long)(cbtot + pb) * oplane + oin) = uint4{0u, 0u, 0u, 0u};
    }
  }
}

// ------------------------------------------------------------------------------------------
// prep_h16_kernel: the same operation for the 16-bit activation stream (16-bit CB8 sources -> 16-bit CB8 output, resample
// SAME / UP2), the form every block-input pass of the 16-bit modes takes.  HBM-bound byte work: 2 B read + 2 B written per
// element (+ 4 B for a per-voxel modulation).  Differences from prep_kernel that matter for the memory system:
//   * a lane keeps its voxel's channel blocks PACKED (4 VGPRs per 8 channels), so up to NB blocks per wave stay resident
//     between the sum-of-squares pass and the apply pass for every channel count of the model family (<= 1280 channels):
//     each source byte is read once;
//   * all of a lane's loads are issued before the first use (NB 16-byte loads in flight per lane);
//   * WS = 1: a wave owns 64 voxels and ALL their channel blocks -- no LDS, no barrier, 256 voxels per workgroup (the fine
//     levels, where the old 64-voxel workgroups lived for a few hundred cycles each);
//     WS = 4: the four waves split the channel blocks of 64 voxels and combine their sums of squares through LDS (coarse
//     levels: few voxels, many channels).
// ------------------------------------------------------------------------------------------
typedef unsigned int pu32x4 __attribute__((ext_vector_type(4)));

template <bool F16>
__device__ __forceinline__ void unpack8(const pu32x4& v, float (&f)[8]) {
  if (F16) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const h2 h = __builtin_bit_cast(h2, (unsigned int)v[j]);
      f[2 * j] = (float)h[0]; f[2 * j + 1] = (float)h[1];
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned int w = v[j];
      f[2 * j] = __uint_as_float(w << 16); f[2 * j + 1] = __uint_as_float(w & 0xffff0000u);
    }
  }
}
template <bool F16>
__device__ __forceinline__ pu32x4 pack8(const float (&f)[8]) {
  pu32x4 r;
  if (F16) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int j = 0; j < 4; ++j) { h2 h; h[0] = (_Float16)f[2 * j]; h[1] = (_Float16)f[2 * j + 1]; r[j] = __builtin_bit_cast(unsigned int, h); }
  } else {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int j = 0; j < 4; ++j) { b2 h; h[0] = (__bf16)f[2 * j]; h[1] = (__bf16)f[2 * j + 1]; r[j] = __builtin_bit_cast(unsigned int, h); }
  }
  return r;
}

template <int WS, int NB, bool F16>
__global__ __launch_bounds__(256) void prep_h16_kernel(PrepArgs pa) {
  const PrepLaunch& L = pa.L;
  __shared__ float red[WS == 4 ? 4 : 1][64];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int S = L.S, Z = L.Z;
  const int vpn = Z * S * S;
  const long vidx = (WS == 4 ? (long)blockIdx.x * 64 : ((long)blockIdx.x * 4 + wv) * 64) + lane;
  const bool valid = vidx < (long)vpn * L.N;
  int n = 0, z = 0, y = 0, x = 0;
  if (valid) {
    n = (int)(vidx / vpn);
    int rem = (int)(vidx - (long)n * vpn);
    z = rem / (S * S); rem -= z * S * S;
    y = rem / S; x = rem - y * S;
  }
  // source plane size and the voxel's coordinates in it: as they are, nearest x2 (UP2), or every second voxel (PICK2);
  // the collage remap (half a patch down / right, into the neighbouring encoder patch) is applied in source coordinates
  const int Ss = L.resample == RS_UP2 ? S / 2 : (L.resample == RS_PICK2 ? 2 * S : S);
  const long splane = (long)Z * Ss * Ss * 8;               // elements per channel block of a source patch
  long soff[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    soff[k] = 0;
    if (k >= L.nsrc) continue;
    int ns = n, ys = y, xs = x;
    if (L.resample == RS_UP2) { ys = y >> 1; xs = x >> 1; }
    if (L.resample == RS_PICK2) { ys = 2 * y; xs = 2 * x; }
    if (L.src[k].collage) {
      const int q1 = L.p1 - 1, q2 = L.p2 - 1;
      const int bi = n / (q1 * q2);
      const int q = n - bi * q1 * q2;
      int i = q / q2, j = q - i * q2;
      ys += Ss / 2; if (ys >= Ss) { ys -= Ss; i += 1; }
      xs += Ss / 2; if (xs >= Ss) { xs -= Ss; j += 1; }
      ns = bi * L.p1 * L.p2 + i * L.p2 + j;
    }
    soff[k] = (long)ns * L.src[k].nstride + ((long)(z * Ss + ys) * Ss + xs) * 8;
  }
  const int cb0 = L.src[0].Cb, cb01 = cb0 + ((L.nsrc > 1) ? L.src[1].Cb : 0);
  const int cbtot = cb01 + ((L.nsrc > 2) ? L.src[2].Cb : 0);
  const int g0 = WS == 4 ? wv : 0;                          // this wave's virtual blocks: g0, g0 + WS, ...
  // Every one of the NB loads is issued UNCONDITIONALLY (a block past the end re-reads the last real block, an invalid lane reads
  // voxel 0's entry: a cache hit, the value is never used): with the loads under `if (gb < cbtot)` the register allocator
  // kept three copies of the block array alive across the branches (234 VGPRs for NB = 16, two waves per SIMD)
  pu32x4 c[NB];
  const int glast = g0 < cbtot ? cbtot - 1 - ((cbtot - 1 - g0) % WS) : 0;      // the wave's last real block (block 0 if it has none)
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int gb = min(g0 + WS * i, glast);
    const int k = (gb >= cb0) + (gb >= cb01);              // wave-uniform
    const int cb = gb - (k == 0 ? 0 : (k == 1 ? cb0 : cb01));
    const uint16_t* sp = (const uint16_t*)(k == 0 ? L.src[0].p : (k == 1 ? L.src[1].p : L.src[2].p));
    const long so = k == 0 ? soff[0] : (k == 1 ? soff[1] : soff[2]);
    c[i] = *(const pu32x4*)(sp + so + (long)cb * splane);
  }
  float rstd = 1.f;
  if (L.norm_w) {
    // Both forms add in ONE order -- four partial sums over the blocks gb = r, r + 4, r + 8, ... (r = 0..3), then
    // p0 + p1 + p2 + p3 -- so that the form (which depends on the size of the call) never changes a result bit: a patch
    // computed inside a large batch equals the same patch computed alone.
    float ps[4] = {0.f, 0.f, 0.f, 0.f};
    if (valid) {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        if (g0 + WS * i < cbtot) {
          float f[8];
          unpack8<F16>(c[i], f);
#pragma unroll
          for (int j = 0; j < 8; ++j) ps[WS == 4 ? 0 : (i & 3)] += f[j] * f[j];
        }
        if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);     // four blocks' floats live at a time, not all NB x 8 of them
      }
    }
    float ssq;
    if (WS == 4) {
      red[wv][lane] = ps[0];
      __syncthreads();
      ssq = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    } else {
      ssq = ps[0] + ps[1] + ps[2] + ps[3];
    }
    rstd = 1.0f / sqrtf(ssq * L.inv_c + TM_EPS);
  }
  if (!valid) return;
  // the apply pass unpacks the resident blocks again: without this the compiler keeps the sum-of-squares pass's unpacked
  // floats alive instead (8 VGPRs per block on top of the 4 packed ones)
#pragma unroll
  for (int i = 0; i < NB; ++i) asm volatile("" : "+v"(c[i]));
  const long oplane = (long)vpn * 8;
  const long oin = ((long)(z * S + y) * S + x) * 8;
  const int img = n / L.per_image;
  uint16_t* const outp = L.out_h + (long)n * L.out_h_nstride + oin;
  uint16_t* const rawp = L.raw_h ? L.raw_h + (long)n * L.raw_h_nstride + oin : nullptr;
  // per-voxel modulation tensors: the output geometry, or (mod_half) half its in-plane resolution
  const long mplane = L.mod_half ? oplane >> 2 : oplane;
  const long mo = (long)n * L.mod_stride + (L.mod_half ? ((long)(z * (S >> 1) + (y >> 1)) * (S >> 1) + (x >> 1)) * 8 : oin);
  constexpr int CH = 4;                                      // blocks per chunk: their modulation loads are issued together
#pragma unroll
  for (int i0 = 0; i0 < NB; i0 += CH) {
    pu32x4 msc[CH], msh[CH];
    if (L.mod == MOD_VOXEL) {
#pragma unroll
      for (int ii = 0; ii < CH; ++ii) {
        const int gb = min(g0 + WS * (i0 + ii), glast);        // unconditional, as the source loads above
        msc[ii] = *(const pu32x4*)(L.mod_scale_h + mo + (long)gb * mplane);
        msh[ii] = *(const pu32x4*)(L.mod_shift_h + mo + (long)gb * mplane);
      }
    } else {
#pragma unroll
      for (int ii = 0; ii < CH; ++ii) { msc[ii] = pu32x4{0u, 0u, 0u, 0u}; msh[ii] = pu32x4{0u, 0u, 0u, 0u}; }
    }
#pragma unroll
    for (int ii = 0; ii < CH; ++ii) {
      const int i = i0 + ii;
      const int gb = g0 + WS * i;
      if (i < NB && gb < cbtot) {
        float v[8];
        unpack8<F16>(c[i], v);
        if (rawp) *(pu32x4*)(rawp + (long)gb * oplane) = c[i];
        if (L.norm_w) {
          const f32x4 w0 = *(const f32x4*)(L.norm_w + gb * 8), w1 = *(const f32x4*)(L.norm_w + gb * 8 + 4);
          const float wn[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = wn[j] * (v[j] * rstd);
        }
        if (L.mod == MOD_IMAGE) {
          const float* scp = L.mod_scale + (long)img * L.mod_stride + gb * 8;
          const float* shp = L.mod_shift + (long)img * L.mod_stride + gb * 8;
          const f32x4 s0 = *(const f32x4*)scp, s1 = *(const f32x4*)(scp + 4), h0 = *(const f32x4*)shp, h1 = *(const f32x4*)(shp + 4);
          const float sc[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
          const float sh[8] = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = v[j] * (1.0f + sc[j]) + sh[j];
        } else if (L.mod == MOD_VOXEL) {
          float sc[8], sh[8];
          unpack8<F16>(msc[ii], sc);
          unpack8<F16>(msh[ii], sh);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = v[j] * (1.0f + sc[j]) + sh[j];
        }
        if (L.act) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = silu_h16(v[j]);
        }
        *(pu32x4*)(outp + (long)gb * oplane) = pack8<F16>(v);
      }
    }
    __builtin_amdgcn_sched_barrier(0);       // one chunk at a time: without it the bf16 build unpacks every block up front (234 VGPRs)
  }
  if (L.pad_blocks && (WS == 1 || wv == 0)) {
    for (int pb = 0; pb < L.pad_blocks; ++pb) {
      *(pu32x4*)(outp + (long)(cbtot + pb) * oplane) = pu32x4{0u, 0u, 0u, 0u};
      if (rawp) *(pu32x4*)(rawp + (long)(cbtot + pb) * oplane) = pu32x4{0u, 0u, 0u, 0u};
    }
  }
}

// Downsample form (ResBlock(down=True), model/MBAblocks.py:254-258, blocks.py:389-403: norm -> SiLU at full resolution, then
// the 2 x 2 average of h and of x): an output voxel gathers its four source voxels, each normalised with its OWN statistics;
// out = mean of the four activated values, raw = mean of the four inputs (the block's residual).  Four waves split the channel
// blocks of 64 output voxels (C <= 256: 8 blocks x 4 sources x 16 bytes resident per lane).
template <int NB, bool F16>
__global__ __launch_bounds__(256) void prep_down_h16_kernel(PrepArgs pa) {
  const PrepLaunch& L = pa.L;
  __shared__ float red[4][4][64];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int S = L.S, Z = L.Z, Ss = 2 * S;
  const int vpn = Z * S * S;
  const long vidx = (long)blockIdx.x * 64 + lane;
  const bool valid = vidx < (long)vpn * L.N;
  int n = 0, z = 0, y = 0, x = 0;
  if (valid) {
    n = (int)(vidx / vpn);
    int rem = (int)(vidx - (long)n * vpn);
    z = rem / (S * S); rem -= z * S * S;
    y = rem / S; x = rem - y * S;
  }
  const long splane = (long)Z * Ss * Ss * 8;
  const uint16_t* sp = (const uint16_t*)L.src[0].p + (long)n * L.src[0].nstride;
  long soff[4];
#pragma unroll
  for (int sub = 0; sub < 4; ++sub) soff[sub] = ((long)(z * Ss + 2 * y + (sub >> 1)) * Ss + 2 * x + (sub & 1)) * 8;
  const int cbtot = L.src[0].Cb;
  pu32x4 c[NB][4];
  float ssq[4] = {0.f, 0.f, 0.f, 0.f};
  if (valid) {
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int gb = wv + 4 * i;
      if (gb < cbtot) {
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) c[i][sub] = *(const pu32x4*)(sp + soff[sub] + (long)gb * splane);
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      if (wv + 4 * i < cbtot) {
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
          float f[8];
          unpack8<F16>(c[i][sub], f);
#pragma unroll
          for (int j = 0; j < 8; ++j) ssq[sub] += f[j] * f[j];
        }
      }
    }
  }
#pragma unroll
  for (int sub = 0; sub < 4; ++sub) red[wv][sub][lane] = ssq[sub];
  __syncthreads();
  if (!valid) return;
  float rstd[4];
#pragma unroll
  for (int sub = 0; sub < 4; ++sub) {
    const float t = red[0][sub][lane] + red[1][sub][lane] + red[2][sub][lane] + red[3][sub][lane];
    rstd[sub] = L.norm_w ? 1.0f / sqrtf(t * L.inv_c + TM_EPS) : 1.0f;
  }
  const long oplane = (long)vpn * 8;
  const long oin = ((long)(z * S + y) * S + x) * 8;
  uint16_t* const outp = L.out_h + (long)n * L.out_h_nstride + oin;
  uint16_t* const rawp = L.raw_h ? L.raw_h + (long)n * L.raw_h_nstride + oin : nullptr;
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int gb = wv + 4 * i;
    if (gb < cbtot) {
      float wn[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) wn[j] = 1.f;
      if (L.norm_w) {
        const f32x4 w0 = *(const f32x4*)(L.norm_w + gb * 8), w1 = *(const f32x4*)(L.norm_w + gb * 8 + 4);
        wn[0] = w0[0]; wn[1] = w0[1]; wn[2] = w0[2]; wn[3] = w0[3]; wn[4] = w1[0]; wn[5] = w1[1]; wn[6] = w1[2]; wn[7] = w1[3];
      }
      float o[8], r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { o[j] = 0.f; r[j] = 0.f; }
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        float v[8];
        unpack8<F16>(c[i][sub], v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          r[j] += v[j];
          float t = L.norm_w ? wn[j] * (v[j] * rstd[sub]) : v[j];
          if (L.act) t = silu_h16(t);
          o[j] += t;
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) { o[j] *= 0.25f; r[j] *= 0.25f; }
      *(pu32x4*)(outp + (long)gb * oplane) = pack8<F16>(o);
      if (rawp) *(pu32x4*)(rawp + (long)gb * oplane) = pack8<F16>(r);
    }
  }
  if (L.pad_blocks && wv == 0) {
    for (int pb = 0; pb < L.pad_blocks; ++pb) {
      *(pu32x4*)(outp + (long)(cbtot + pb) * oplane) = pu32x4{0u, 0u, 0u, 0u};
      if (rawp) *(pu32x4*)(rawp + (long)(cbtot + pb) * oplane) = pu32x4{0u, 0u, 0u, 0u};
    }
  }
}

// variant: 0 = automatic, 1 = prep_kernel (the generic form), 2 = prep_h16_kernel with one wave per 64 voxels, 3 = prep_h16_kernel
// with the four waves of a workgroup splitting the channel blocks
static int g_prep_variant = 0;
void set_prep_variant(int v) { g_prep_variant = v; }

template <bool F16>
static bool launch_prep_h16(const PrepLaunch& L, hipStream_t s, int variant) {
  PrepArgs pa; pa.L = L;
  const long vox = (long)L.N * L.Z * L.S * L.S;
  int cbtot = 0;
  for (int k = 0; k < L.nsrc; ++k) cbtot += L.src[k].Cb;
  // measured on the test_brn tile shapes (tools/bench_prep.py, profiles/r02_bench_prep.txt): the one-wave form wins for up to 12
  // channel blocks on the fine levels, the split form everywhere else
  bool one = cbtot <= 12 && vox / 256 >= 2048;
  if (variant == 2) { if (cbtot > 32) return false; one = true; }
  if (variant == 3) one = false;
  if (one) {
    const unsigned grid = (unsigned)((vox + 255) / 256);
    if (cbtot <= 8) hipLaunchKernelGGL((prep_h16_kernel<1, 8, F16>), dim3(grid), dim3(256), 0, s, pa);
    else if (cbtot <= 16) hipLaunchKernelGGL((prep_h16_kernel<1, 16, F16>), dim3(grid), dim3(256), 0, s, pa);
    else hipLaunchKernelGGL((prep_h16_kernel<1, 32, F16>), dim3(grid), dim3(256), 0, s, pa);
  } else {
    if (cbtot > 160) return false;
    const unsigned grid = (unsigned)((vox + 63) / 64);
    if (cbtot <= 32) hipLaunchKernelGGL((prep_h16_kernel<4, 8, F16>), dim3(grid), dim3(256), 0, s, pa);
    else if (cbtot <= 64) hipLaunchKernelGGL((prep_h16_kernel<4, 16, F16>), dim3(grid), dim3(256), 0, s, pa);
    else hipLaunchKernelGGL((prep_h16_kernel<4, 40, F16>), dim3(grid), dim3(256), 0, s, pa);
  }
  return true;
}

hipError_t launch_prep(const PrepLaunch& L, hipStream_t s) {
  static const int env_form = getenv("TM_PREP_FORM") ? atoi(getenv("TM_PREP_FORM")) : 0;       // A/B timing only
  const int form = g_prep_variant ? g_prep_variant : env_form;
  if (form != 1 && L.src_h && L.out_h && !L.out && !L.raw && !L.drop_mask && L.resample == RS_DOWN2 && L.nsrc == 1 &&
      !L.src[0].collage && L.mod == MOD_NONE && L.src[0].Cb <= 32) {
    PrepArgs pa; pa.L = L;
    const unsigned grid = (unsigned)(((long)L.N * L.Z * L.S * L.S + 63) / 64);
    if (L.h_f16) hipLaunchKernelGGL((prep_down_h16_kernel<8, true>), dim3(grid), dim3(256), 0, s, pa);
    else hipLaunchKernelGGL((prep_down_h16_kernel<8, false>), dim3(grid), dim3(256), 0, s, pa);
    return hipGetLastError();
  }
  if (form != 1 && L.src_h && L.out_h && !L.out && !L.raw && !L.drop_mask && L.resample != RS_DOWN2 &&
      (L.mod != MOD_VOXEL || L.mod_scale_h)) {
    if (L.h_f16 ? launch_prep_h16<true>(L, s, form) : launch_prep_h16<false>(L, s, form)) return hipGetLastError();
  }

  PrepArgs pa; pa.L = L;
  const long vox = (long)L.N * L.Z * L.S * L.S;
  const unsigned grid = (unsigned)((vox + 63) / 64);
  int cbtot = 0;
  for (int k = 0; k < L.nsrc; ++k) cbtot += L.src[k].Cb;
  if (L.resample == RS_DOWN2) hipLaunchKernelGGL((prep_kernel<4, false>), dim3(grid), dim3(256), 0, s, pa);
  else if (cbtot <= 32) hipLaunchKernelGGL((prep_kernel<1, true>), dim3(grid), dim3(256), 0, s, pa);
  else hipLaunchKernelGGL((prep_kernel<1, false>), dim3(grid), dim3(256), 0, s, pa);
  return hipGetLastError();
}

// ==========================================================================================
// Generic direct Conv3d on VALU (fp32 VALU rate == fp32 MFMA rate on gfx950, and these
// layers are < 1 % of the FLOPs): stem Conv3d(2->64,(1,3,3)) (model/unet_ours.py:110-114),
// head Conv3d(64->2,(1,3,3)) (:274-275), RNA pyramid SiLU->Conv3d(1,3,3)->Upsample
// (:290-295) and down_z Conv3d(G->G,(ker,3,3),pad (0,1,1)) (model/MBAblocks.py:472-474).
// lane = output voxel, blockIdx.y = block of 8 couts: weight reads are wave-uniform.
// weights are pre-transposed to [tap][Cin][Cout_pad8].
// ==========================================================================================
struct DirectArgs { DirectLaunch L; int Cop; };

__global__ __launch_bounds__(256) void conv_direct_kernel(DirectArgs da) {
  const DirectLaunch& L = da.L;
  const int S = L.S;
  const long vpn = (long)L.Zout * S * S;
  const long vidx = (long)blockIdx.x * 256 + threadIdx.x;
  if (vidx >= vpn * L.N) return;
  const int n = (int)(vidx / vpn);
  int rem = (int)(vidx - (long)n * vpn);
  const int zo = rem / (S * S); rem -= zo * S * S;
  const int y = rem / S, x = rem - y * S;
  const int cob = blockIdx.y;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  const float* xb = L.x + (long)n * L.ax.sN;
  for (int kz = 0; kz < L.kz; ++kz) {
    const int zi = zo + kz - L.pz;
    if (zi < 0 || zi >= L.Zin) continue;
    for (int ky = 0; ky < L.ky; ++ky) {
      const int yi = y + ky - L.py;
      for (int kx = 0; kx < L.kx; ++kx) {
        const int xi = x + kx - L.px;
        const bool ok = yi >= 0 && yi < S && xi >= 0 && xi < S;
        const long base = zi * L.ax.sZ + (long)yi * L.ax.sY + (long)xi * L.ax.sX;
        const int tap = (kz * L.ky + ky) * L.kx + kx;
        const float* wt = L.w + ((long)tap * L.Cin) * da.Cop + cob * 8;
        for (int ci = 0; ci < L.Cin; ++ci) {
          float xv = 0.f;
          if (ok) xv = xb[base + (long)(ci >> 3) * L.ax.sCb + (long)(ci & 7) * L.ax.sC8];
          if (L.silu_in) xv = silu_f(xv);
          const float* wp = wt + (long)ci * da.Cop;
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = fmaf(wp[j], xv, acc[j]);
        }
      }
    }
  }
  float* yb = L.y + (long)n * L.ay.sN + (long)cob * L.ay.sCb;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int co = cob * 8 + j;
    if (co >= L.Cout) break;
    const float v = acc[j] + L.bias[co];
    if (L.up2_out) {
      const long o = zo * L.ay.sZ + (long)(2 * y) * L.ay.sY + (long)(2 * x) * L.ay.sX + (long)j * L.ay.sC8;
      yb[o] = v; yb[o + L.ay.sX] = v; yb[o + L.ay.sY] = v; yb[o + L.ay.sY + L.ay.sX] = v;
    } else {
      yb[zo * L.ay.sZ + (long)y * L.ay.sY + (long)x * L.ay.sX + (long)j * L.ay.sC8] = v;
    }
  }
}

Acc5 acc_ncdhw(int C, int Z, int H, int W) {
  Acc5 a;
  a.sX = 1; a.sY = W; a.sZ = (long)H * W; a.sC8 = (long)Z * H * W; a.sCb = 8 * a.sC8;
  a.sN = (long)C * Z * H * W;
  return a;
}
Acc5 acc_cb8(const TV& t) {
  Acc5 a;
  a.sC8 = 1; a.sX = 8; a.sY = (long)t.W * 8; a.sZ = (long)t.H * t.W * 8; a.sCb = t.plane();
  a.sN = t.nstride;
  return a;
}

hipError_t launch_conv_direct(const DirectLaunch& L, hipStream_t s) {
  DirectArgs da; da.L = L; da.Cop = (L.Cout + 7) / 8 * 8;
  const long vox = (long)L.N * L.Zout * L.S * L.S;
  dim3 grid((unsigned)((vox + 255) / 256), (unsigned)(da.Cop / 8));
  hipLaunchKernelGGL(conv_direct_kernel, grid, dim3(256), 0, s, da);
  return hipGetLastError();
}

// ==========================================================================================
// Stem Conv3d(n_stain -> 64k, (1,3,3)) straight from the NCHW '(s z) h w' state into CB8
// (model/unet_ours.py:110-114,377) and head Conv3d(64 -> n_stain, (1,3,3)) from CB8 straight
// into the NCHW eps tensor (:274-275,424).  lane = voxel; weights are wave-uniform (scalar loads).
// ==========================================================================================
struct StemArgs {
  const float* x; float* y; const float* w; const float* bias;   // w: [tap 9][ci][Cop]
  int N, Cin, Cout, Cop, Z, S;
  long y_nstride;
  uint16_t* y_h; long yh_nstride; int h_f16;                     // 16-bit CB8 output instead of y (bf16, or fp16 with h_f16)
};
__global__ __launch_bounds__(256) void stem_kernel(StemArgs a) {
  const int S = a.S;
  const long vpn = (long)a.Z * S * S;
  const long vidx = (long)blockIdx.x * 256 + threadIdx.x;
  if (vidx >= vpn * a.N) return;
  const int n = (int)(vidx / vpn);
  int rem = (int)(vidx - (long)n * vpn);
  const int z = rem / (S * S); rem -= z * S * S;
  const int y = rem / S, x = rem - y * S;
  const int cob0 = blockIdx.y * 4;                 // 4 cout blocks (32 couts) per thread
  // accumulators as pairs: one v_pk_fma_f32 per two couts (the weights are wave-uniform SGPR pairs); each accumulator still sees
  // the same sequence of FMAs, so the result bits do not change
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  f32x2_t acc2[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc2[j] = f32x2_t{a.bias[cob0 * 8 + 2 * j], a.bias[cob0 * 8 + 2 * j + 1]};
  for (int ci = 0; ci < a.Cin; ++ci) {
    const float* xp = a.x + (((long)n * a.Cin + ci) * a.Z + z) * S * S;
    // one tap's 32 wave-uniform weights (SGPRs) at a time: with the nine taps unrolled the 288 weights of a channel did not
    // fit the scalar register file and were spilled to VGPR lanes (v_writelane / v_readlane: 3 x the FMA count)
#pragma unroll 1
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll 1
      for (int kx = 0; kx < 3; ++kx) {
        const int yi = y + ky - 1, xi = x + kx - 1;
        const float xv = (yi >= 0 && yi < S && xi >= 0 && xi < S) ? xp[yi * S + xi] : 0.f;
        const float* wp = a.w + ((long)(ky * 3 + kx) * a.Cin + ci) * a.Cop + cob0 * 8;
        const f32x2_t xv2 = {xv, xv};
#pragma unroll
        for (int j = 0; j < 16; ++j) acc2[j] = __builtin_elementwise_fma(f32x2_t{wp[2 * j], wp[2 * j + 1]}, xv2, acc2[j]);
      }
    }
  }
  float acc[32];
#pragma unroll
  for (int j = 0; j < 16; ++j) { acc[2 * j] = acc2[j][0]; acc[2 * j + 1] = acc2[j][1]; }
  if (a.y_h) {
    uint16_t* hp = a.y_h + (long)n * a.yh_nstride + ((long)(z * S + y) * S + x) * 8;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      uint16_t* p = hp + (long)(cob0 + b) * vpn * 8;
      if (a.h_f16) {
        typedef _Float16 f16x8_o __attribute__((ext_vector_type(8)));
        f16x8_o o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (_Float16)acc[b * 8 + j];
        *(f16x8_o*)p = o;
      } else {
        typedef __bf16 bf16x8_o __attribute__((ext_vector_type(8)));
        bf16x8_o o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (__bf16)acc[b * 8 + j];
        *(bf16x8_o*)p = o;
      }
    }
    return;
  }
  float* yp = a.y + (long)n * a.y_nstride + ((long)(z * S + y) * S + x) * 8;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    float* p = yp + (long)(cob0 + b) * vpn * 8;
    *(f32x4*)p = f32x4{acc[b * 8 + 0], acc[b * 8 + 1], acc[b * 8 + 2], acc[b * 8 + 3]};
    *(f32x4*)(p + 4) = f32x4{acc[b * 8 + 4], acc[b * 8 + 5], acc[b * 8 + 6], acc[b * 8 + 7]};
  }
}
hipError_t launch_stem(const float* x, TV y, const float* w, const float* bias, int Cin, hipStream_t s, uint16_t* y_h,
                       long yh_nstride, int h_f16) {
  if (y.C % 32) return hipErrorInvalidValue;
  StemArgs a{x, y.p, w, bias, y.N, Cin, y.C, y.Cb * 8, y.Z, y.H, y.nstride, y_h, yh_nstride, h_f16};
  const long vox = (long)y.N * y.Z * y.H * y.W;
  hipLaunchKernelGGL(stem_kernel, dim3((unsigned)((vox + 255) / 256), (unsigned)(y.C / 32)), dim3(256), 0, s, a);
  return hipGetLastError();
}

struct HeadArgs {
  const float* x; long x_nstride; int Cb;
  float* y; const float* w; const float* bias;                    // w: [tap 9][ci][8]
  int N, Cout, Z, S;
};
// H16: 0 = fp32 CB8 input, 1 = bf16, 2 = fp16 (the 16-bit modes hand the normalised, activated tensor over in 16 bits like
// every other conv input: half the bytes of the 9-tap gather)
// CO = couts computed (4: the two-stain models -- half the FMAs of the padded 8; 8: anything up to 8)
template <int H16, int CO>
__global__ __launch_bounds__(256) void head_kernel(HeadArgs a) {
  const int S = a.S;
  const long vpn = (long)a.Z * S * S;
  const long vidx = (long)blockIdx.x * 256 + threadIdx.x;
  if (vidx >= vpn * a.N) return;
  const int n = (int)(vidx / vpn);
  int rem = (int)(vidx - (long)n * vpn);
  const int z = rem / (S * S); rem -= z * S * S;
  const int y = rem / S, x = rem - y * S;
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  f32x2_t acc2[CO / 2];                                            // pairs of couts: v_pk_fma_f32, same FMA sequence per cout
#pragma unroll
  for (int j = 0; j < CO / 2; ++j) acc2[j] = f32x2_t{0.f, 0.f};
  const float* xb = a.x + (long)n * a.x_nstride;                   // H16: the same offsets count 16-bit elements
  const uint16_t* xbh = (const uint16_t*)a.x + (long)n * a.x_nstride;
  // channel block outermost: the 9 taps of one block touch 3 rows of one plane back to back (L1 hits); with the
  // taps outermost every tap strides through all Cb planes and the rows are evicted before the next tap returns
  // to them (measured: 8.7x the input bytes fetched from HBM)
  for (int cb = 0; cb < a.Cb; ++cb) {
    const long co = (long)cb * vpn * 8;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int yi = y + ky - 1, xi = x + kx - 1;
        if (yi < 0 || yi >= S || xi < 0 || xi >= S) continue;
        const long eo = co + ((long)(z * S + yi) * S + xi) * 8;
        float xv[8];
        if (H16 == 0) {
          const f32x4 a0 = *(const f32x4*)(xb + eo), a1 = *(const f32x4*)(xb + eo + 4);
          xv[0] = a0[0]; xv[1] = a0[1]; xv[2] = a0[2]; xv[3] = a0[3]; xv[4] = a1[0]; xv[5] = a1[1]; xv[6] = a1[2]; xv[7] = a1[3];
        } else if (H16 == 1) {
          typedef __bf16 bf16x8_i __attribute__((ext_vector_type(8)));
          const bf16x8_i v = *(const bf16x8_i*)(xbh + eo);
#pragma unroll
          for (int c = 0; c < 8; ++c) xv[c] = (float)v[c];
        } else {
          typedef _Float16 f16x8_i __attribute__((ext_vector_type(8)));
          const f16x8_i v = *(const f16x8_i*)(xbh + eo);
#pragma unroll
          for (int c = 0; c < 8; ++c) xv[c] = (float)v[c];
        }
        const float* wp = a.w + ((long)(ky * 3 + kx) * a.Cb + cb) * 64;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const f32x2_t xv2 = {xv[c], xv[c]};
#pragma unroll
          for (int j = 0; j < CO / 2; ++j)
            acc2[j] = __builtin_elementwise_fma(f32x2_t{wp[c * 8 + 2 * j], wp[c * 8 + 2 * j + 1]}, xv2, acc2[j]);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < CO; ++j)
    if (j < a.Cout) a.y[(((long)n * a.Cout + j) * a.Z + z) * S * S + (long)y * S + x] = acc2[j / 2][j % 2] + a.bias[j];
}
hipError_t launch_head(TV x, float* y, const float* w, const float* bias, int Cout, hipStream_t s, int h16) {
  if (Cout > 8 || h16 < 0 || h16 > 2) return hipErrorInvalidValue;
  HeadArgs a{x.p, x.nstride, x.Cb, y, w, bias, x.N, Cout, x.Z, x.H};
  const long vox = (long)x.N * x.Z * x.H * x.W;
  const dim3 grid((unsigned)((vox + 255) / 256));
  if (Cout <= 4) {
    if (h16 == 0) hipLaunchKernelGGL((head_kernel<0, 4>), grid, dim3(256), 0, s, a);
    else if (h16 == 1) hipLaunchKernelGGL((head_kernel<1, 4>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((head_kernel<2, 4>), grid, dim3(256), 0, s, a);
  } else {
    if (h16 == 0) hipLaunchKernelGGL((head_kernel<0, 8>), grid, dim3(256), 0, s, a);
    else if (h16 == 1) hipLaunchKernelGGL((head_kernel<1, 8>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((head_kernel<2, 8>), grid, dim3(256), 0, s, a);
  }
  return hipGetLastError();
}

// ==========================================================================================
// layout converters
// ==========================================================================================
__global__ void to_cb8_kernel(const float* x, float* y, int N, int C, int Cb, long vpn, long y_nstride) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;     // over N*Cb*vpn*8
  const long tot = (long)N * Cb * vpn * 8;
  if (i >= tot) return;
  const int c8 = (int)(i & 7);
  long r = i >> 3;
  const long v = r % vpn; r /= vpn;
  const int cb = (int)(r % Cb);
  const int n = (int)(r / Cb);
  const int c = cb * 8 + c8;
  y[(long)n * y_nstride + ((long)cb * vpn + v) * 8 + c8] = (c < C) ? x[((long)n * C + c) * vpn + v] : 0.f;
}
__global__ void from_cb8_kernel(const float* x, float* y, int N, int C, long vpn, long x_nstride) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;     // over N*C*vpn
  const long tot = (long)N * C * vpn;
  if (i >= tot) return;
  const long v = i % vpn;
  long r = i / vpn;
  const int c = (int)(r % C);
  const int n = (int)(r / C);
  y[i] = x[(long)n * x_nstride + ((long)(c >> 3) * vpn + v) * 8 + (c & 7)];
}
hipError_t launch_to_cb8(const float* x, TV y, hipStream_t s) {
  const long vpn = (long)y.Z * y.H * y.W;
  const long tot = (long)y.N * y.Cb * vpn * 8;
  hipLaunchKernelGGL(to_cb8_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, x, y.p, y.N, y.C, y.Cb, vpn, y.nstride);
  return hipGetLastError();
}
hipError_t launch_from_cb8(TV x, float* y, hipStream_t s) {
  const long vpn = (long)x.Z * x.H * x.W;
  const long tot = (long)x.N * x.C * vpn;
  hipLaunchKernelGGL(from_cb8_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, x.p, y, x.N, x.C, vpn, x.nstride);
  return hipGetLastError();
}

// ==========================================================================================
// timestep embedding: sinusoid -> Linear -> SiLU -> Linear  (model/nn.py:187-206,
// model/unet_ours.py:442-476), then every ResBlock's emb_layers = SiLU -> Linear(E, 2*Cout)
// (model/MBAblocks.py:168-171) as ONE matrix [sum 2*Cout][E]: the value depends only on t,
// so it is computed once per image instead of once per patch per block.
// ==========================================================================================
__global__ __launch_bounds__(256) void time_embed_kernel(const int64_t* t, int ch, int E, const float* w1,
                                                         const float* b1, const float* w2, const float* b2,
                                                         float* te) {
  extern __shared__ float sm[];       // [ch] sinusoid, [E] hidden
  float* sinu = sm;
  float* hid = sm + ch;
  const int b = blockIdx.x, tid = threadIdx.x;
  const float tv = (float)t[b];
  const int half = ch / 2;
  for (int i = tid; i < ch; i += 256) {
    const int k = (i < half) ? i : i - half;
    const float fr = expf(-logf(10000.0f) * (float)k / (float)half);
    const float arg = tv * fr;
    sinu[i] = (i < half) ? cosf(arg) : sinf(arg);
  }
  __syncthreads();
  for (int o = tid; o < E; o += 256) {
    float acc = b1[o];
    for (int k = 0; k < ch; ++k) acc = fmaf(w1[(long)o * ch + k], sinu[k], acc);
    hid[o] = silu_f(acc);
  }
  __syncthreads();
  for (int o = tid; o < E; o += 256) {
    float acc = b2[o];
    for (int k = 0; k < E; ++k) acc = fmaf(w2[(long)o * E + k], hid[k], acc);
    te[(long)b * E + o] = acc;
  }
}
hipError_t launch_time_embed(const int64_t* t, int b, int ch, int E, const float* w1, const float* b1,
                             const float* w2, const float* b2, float* te, hipStream_t s) {
  hipLaunchKernelGGL(time_embed_kernel, dim3(b), dim3(256), (ch + E) * sizeof(float), s, t, ch, E, w1, b1, w2, b2, te);
  return hipGetLastError();
}

// one wave per output row e; lanes split E; weight row kept in registers across images
__global__ __launch_bounds__(256) void emb_all_kernel(const float* te, int b, int E, const float* wall,
                                                      const float* ball, int tot, float* ss) {
  const int lane = threadIdx.x & 63;
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= tot) return;
  float w[16];
  const int per = E / 64;            // E <= 1024
#pragma unroll
  for (int j = 0; j < 16; ++j) w[j] = (j < per) ? wall[(long)e * E + j * 64 + lane] : 0.f;
  const float bias = ball[e];
  for (int i = 0; i < b; ++i) {
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (j < per) acc = fmaf(w[j], silu_f(te[(long)i * E + j * 64 + lane]), acc);
    acc = wave_sum(acc);
    if (lane == 0) ss[(long)i * tot + e] = acc + bias;
  }
}
hipError_t launch_emb_all(const float* te, int b, int E, const float* wall, const float* ball, int tot,
                          float* ss, hipStream_t s) {
  if (E % 64 || E > 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(emb_all_kernel, dim3((tot + 3) / 4), dim3(256), 0, s, te, b, E, wall, ball, tot, ss);
  return hipGetLastError();
}

}  // namespace tmk
