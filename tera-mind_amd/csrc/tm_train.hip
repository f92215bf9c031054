// Training slice: backward kernels of one ResBlock (model/MBAblocks.py:237-299,302-368) --
//   prep_bwd_kernel    backward of  y = Dropout(SiLU(RMSNorm_C(x) * w * (1 + scale) + shift))  (in_layers[0:2], out_layers[0:3])
//   conv_wgrad_kernel  dW of Conv3d(k = 3x3x3 pad 1, Z == 2) and of the 1x1x1 skip conv
//   chan_sum_kernel    bias gradients (sum over voxels per channel)
// The data gradient of the convs (dgrad) needs no kernel of its own: a stride-1 "same" conv's dgrad is the forward conv of
// dY with the kernel flipped in every axis and cin <-> cout transposed, so it runs on conv3d_mfma / conv1_mfma with weights
// re-packed by the host (tm_op_conv_dgrad in tm_model.hip).
// fp32 throughout (training in the reference is fp16-mixed on top of fp32 master weights; the slice checks gradients
// against torch.autograd of the fp32 oracle).  Layout: CB8 fp32 [N][Cb][Z][H][W][8] as everywhere.
#include "tm_device.h"

namespace tmk {

// ------------------------------------------------------------------------------------------------------------------
// prep backward.  Forward (per voxel v, channel c; img = patch / per_image):
//   xh = x * rstd(v),  rstd = rsqrt(mean_c x^2 + eps)          LlamaRMSNorm(dim=1), MBAblocks.py:21-43
//   n  = xh * w[c]
//   m  = n * (1 + scale[img][c]) + shift[img][c]               apply_conditions, MBAblocks.py:356-367 (optional)
//   s  = SiLU(m);  y = s * mask * drop_scale                   nn.SiLU, nn.Dropout(p) with a SUPPLIED keep mask (optional)
// Backward, g = dL/dy:
//   ds = g * mask * drop_scale;  dm = ds * sig(m) * (1 + m * (1 - sig(m)))
//   dscale[img][c] += dm * n;  dshift[img][c] += dm;  dn = dm * (1 + scale)
//   dw[c] += dn * xh;  dxh = dn * w
//   dx = rstd * (dxh - xh * mean_c(dxh * xh))
// lane = voxel, the workgroup's 4 waves split the channel blocks (as prep_kernel); two passes over the channels
// (first: mean_c(dxh * xh), second: dx).  The per-channel sums (dw, dscale, dshift) are reduced in TWO STAGES so that the
// gradients are bitwise reproducible: the 64 voxels of a workgroup in registers (wave_sum), one partial per (workgroup,
// channel) STORED to a scratch slab, and prep_bwd_reduce_kernel adds the slabs of the workgroups in index order -- no float
// atomics (their sum depends on arrival order: the last bits changed from run to run).
// ------------------------------------------------------------------------------------------------------------------
struct PrepBwdArgs {
  const float* x; long x_ns;            // forward input (pre-norm), CB8 with Cb blocks
  const float* g; long g_ns;            // dL/dy, CB8
  const float* mask; long mask_ns;      // keep mask (0 / 1) CB8 or null
  float drop_scale;                     // 1 / (1 - p)
  const float* w;                       // [Cb*8] norm weight (zero in the pad slots)
  const float* scale; const float* shift; long mod_stride; int per_image;   // [img][..] or null
  float* dx; long dx_ns;
  float* part_dw;                       // [workgroups][Cb*8] partial sums of this workgroup's 64 voxels
  float* part_ds; float* part_dh;       // [workgroups][2 (image of lane 0 | the next image)][Cb*8] partial dscale / dshift, or null
  int N, Cb, Z, S; float inv_c;
};

__global__ __launch_bounds__(256) void prep_bwd_kernel(PrepBwdArgs a) {
  __shared__ float red[4][64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const long vpn = (long)a.Z * a.S * a.S;
  const long vidx = (long)blockIdx.x * 64 + lane;
  const bool valid = vidx < vpn * a.N;
  const int n = valid ? (int)(vidx / vpn) : 0;
  const long off = valid ? (vidx - (long)n * vpn) * 8 : 0;
  const int img = n / a.per_image;
  const long plane = vpn * 8;
  // pass 0: rstd
  float ssq = 0.f;
  if (valid)
    for (int cb = wv; cb < a.Cb; cb += 4) {
      const float* p = a.x + (long)n * a.x_ns + (long)cb * plane + off;
      const f32x4 v0 = *(const f32x4*)p, v1 = *(const f32x4*)(p + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) ssq += v0[j] * v0[j] + v1[j] * v1[j];
    }
  red[wv][lane] = ssq;
  __syncthreads();
  const float rstd = 1.0f / sqrtf((red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]) * a.inv_c + TM_EPS);
  __syncthreads();
  // dxh for one channel block (recomputed in both passes: cheaper than keeping Cb * 8 values per lane)
  auto block = [&](int cb, float (&xh)[8], float (&dxh)[8], float (&dm)[8], float (&nn)[8]) {
    const float* p = a.x + (long)n * a.x_ns + (long)cb * plane + off;
    const float* gp = a.g + (long)n * a.g_ns + (long)cb * plane + off;
    const f32x4 v0 = *(const f32x4*)p, v1 = *(const f32x4*)(p + 4);
    const f32x4 g0 = *(const f32x4*)gp, g1 = *(const f32x4*)(gp + 4);
    f32x4 k0 = {1.f, 1.f, 1.f, 1.f}, k1 = {1.f, 1.f, 1.f, 1.f};
    if (a.mask) { const float* mp = a.mask + (long)n * a.mask_ns + (long)cb * plane + off; k0 = *(const f32x4*)mp; k1 = *(const f32x4*)(mp + 4); }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = cb * 8 + j;
      const float xv = j < 4 ? v0[j] : v1[j - 4], gv = j < 4 ? g0[j] : g1[j - 4], kv = j < 4 ? k0[j] : k1[j - 4];
      const float wc = a.w[c];
      const float sc = a.scale ? a.scale[(long)img * a.mod_stride + c] : 0.f;
      const float sh = a.shift ? a.shift[(long)img * a.mod_stride + c] : 0.f;
      xh[j] = xv * rstd;
      nn[j] = xh[j] * wc;
      const float mm = nn[j] * (1.0f + sc) + sh;
      const float sg = 1.0f / (1.0f + expf(-mm));
      const float ds = gv * kv * a.drop_scale;
      dm[j] = ds * sg * (1.0f + mm * (1.0f - sg));
      dxh[j] = dm[j] * (1.0f + sc) * wc;
    }
  };
  // pass 1: mean_c(dxh * xh) per voxel, and the per-channel sums
  float dot = 0.f;
  for (int cb = wv; cb < a.Cb; cb += 4) {
    float xh[8], dxh[8], dm[8], nn[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { xh[j] = 0.f; dxh[j] = 0.f; dm[j] = 0.f; nn[j] = 0.f; }
    if (valid) block(cb, xh, dxh, dm, nn);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = cb * 8 + j;
      dot += dxh[j] * xh[j];
      // dw[c] = sum_v dn * xh with dn = dm * (1 + scale): dxh * xh = dn * w * xh, so divide the weight back out is
      // avoided by summing dn * xh directly
      const float sc = (a.scale && valid) ? a.scale[(long)img * a.mod_stride + c] : 0.f;
      const float dwv = wave_sum(dm[j] * (1.0f + sc) * xh[j]);
      if (lane == 0) a.part_dw[(long)blockIdx.x * a.Cb * 8 + c] = dwv;
    }
    if (a.part_ds) {
      // the 64 voxels of a wave may belong to two images only at an image boundary: reduce per image of lane 0 and of
      // the last lane (per_image * vpn >= 64 for every geometry of the model: one patch has >= 128 voxels)
      const int img_lo = __shfl(img, 0, 64);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = cb * 8 + j;
        const float s_lo = wave_sum(img == img_lo ? dm[j] * nn[j] : 0.f), h_lo = wave_sum(img == img_lo ? dm[j] : 0.f);
        const float s_hi = wave_sum(img != img_lo ? dm[j] * nn[j] : 0.f), h_hi = wave_sum(img != img_lo ? dm[j] : 0.f);
        if (lane == 0) {
          const long pb = (long)blockIdx.x * 2 * a.Cb * 8;
          a.part_ds[pb + c] = s_lo; a.part_dh[pb + c] = h_lo;
          a.part_ds[pb + a.Cb * 8 + c] = s_hi; a.part_dh[pb + a.Cb * 8 + c] = h_hi;
        }
      }
    }
  }
  red[wv][lane] = dot;
  __syncthreads();
  const float mean_dot = (red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]) * a.inv_c;
  if (!valid) return;
  // pass 2: dx
  for (int cb = wv; cb < a.Cb; cb += 4) {
    float xh[8], dxh[8], dm[8], nn[8];
    block(cb, xh, dxh, dm, nn);
    float* dp = a.dx + (long)n * a.dx_ns + (long)cb * plane + off;
    f32x4 o0, o1;
#pragma unroll
    for (int j = 0; j < 4; ++j) { o0[j] = rstd * (dxh[j] - xh[j] * mean_dot); o1[j] = rstd * (dxh[4 + j] - xh[4 + j] * mean_dot); }
    *(f32x4*)dp = o0;
    *(f32x4*)(dp + 4) = o1;
  }
}

// stage 2: out[c] = sum over workgroups (in index order) of part[wg][c]; one thread per channel, eight independent chains
// combined in a fixed order
__global__ __launch_bounds__(64) void prep_bwd_reduce_dw_kernel(const float* part, long nwg, int C8, float* dw) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C8) return;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  long wg = 0;
  for (; wg + 8 <= nwg; wg += 8)
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] += part[(wg + u) * C8 + c];
  for (int u = 0; wg < nwg; ++wg, ++u) acc[u] += part[wg * C8 + c];
  dw[c] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
}
// dscale / dshift[img][c]: the workgroups that hold voxels of image `img` form the index range [img V / 64, ((img + 1) V - 1) / 64]
// (V = per_image * voxels per patch >= 64): a workgroup whose first voxel belongs to `img` contributes its first slab, the
// one workgroup that starts in image img - 1 and ends in img its second slab
__global__ __launch_bounds__(64) void prep_bwd_reduce_mod_kernel(const float* part_ds, const float* part_dh, long nwg, long V, int C8,
                                                                 long mod_stride, float* dscale, float* dshift) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  const long img = blockIdx.y;
  if (c >= C8) return;
  const long wa = img * V / 64, wb = ((img + 1) * V - 1) / 64;
  float s_ = 0.f, h_ = 0.f;
  for (long wg = wa; wg <= wb && wg < nwg; ++wg) {
    const long lo = wg * 64 / V;                       // image of the workgroup's first voxel
    const long pb = (wg * 2 + (lo == img ? 0 : 1)) * C8 + c;
    s_ += part_ds[pb];
    h_ += part_dh[pb];
  }
  dscale[img * mod_stride + c] = s_;
  dshift[img * mod_stride + c] = h_;
}

size_t prep_bwd_scratch_floats(int N, int Cb, int Z, int S, bool with_mod) {
  const long nwg = ((long)N * Z * S * S + 63) / 64;
  return (size_t)nwg * Cb * 8 * (with_mod ? 5 : 1);
}

hipError_t launch_prep_bwd(const float* x, long x_ns, const float* g, long g_ns, const float* mask, long mask_ns, float drop_scale,
                           const float* w, const float* scale, const float* shift, long mod_stride, int per_image, float* dx,
                           long dx_ns, float* dw, float* dscale, float* dshift, int N, int Cb, int C_real, int Z, int S,
                           float* scratch, hipStream_t s) {
  if (per_image < 1 || (long)per_image * Z * S * S < 64 || !scratch) return hipErrorInvalidValue;
  const long vox = (long)N * Z * S * S, nwg = (vox + 63) / 64;
  const int C8 = Cb * 8;
  float* part_dw = scratch;
  float* part_ds = scale ? scratch + nwg * C8 : nullptr;
  float* part_dh = scale ? part_ds + nwg * 2 * C8 : nullptr;
  PrepBwdArgs a{x, x_ns, g, g_ns, mask, mask_ns, drop_scale, w, scale, shift, mod_stride, per_image, dx, dx_ns, part_dw, part_ds, part_dh,
                N, Cb, Z, S, 1.0f / (float)C_real};
  hipLaunchKernelGGL(prep_bwd_kernel, dim3((unsigned)nwg), dim3(256), 0, s, a);
  hipLaunchKernelGGL(prep_bwd_reduce_dw_kernel, dim3((unsigned)((C8 + 63) / 64)), dim3(64), 0, s, part_dw, nwg, C8, dw);
  if (scale) {
    const long nimg = (N + per_image - 1) / per_image;
    hipLaunchKernelGGL(prep_bwd_reduce_mod_kernel, dim3((unsigned)((C8 + 63) / 64), (unsigned)nimg), dim3(64), 0, s, part_ds, part_dh, nwg,
                       (long)per_image * Z * S * S, C8, mod_stride, dscale, dshift);
  }
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// Conv3d weight gradient:  dW[co][ci][kz][ky][kx] = sum_{n,z,y,x} dY[n][co][z][y][x] * X[n][ci][z+kz-pz][y+ky-1][x+kx-1]
// (taps 27: 3x3x3 pad 1 on Z planes; taps 1: 1x1x1).  One workgroup per (cout block of 8, cin block of 8): it walks
// every patch in 8 x 8 x Z voxel tiles, stages the dY tile and the halo'd X tile of its two channel blocks in LDS and
// accumulates.  Thread (co, ci, q): q = wave splits the taps (tap t belongs to wave t % 4).  The result is written in the
// reference's parameter layout [Cout][Cin][taps].
// ------------------------------------------------------------------------------------------------------------------
struct WgradArgs {
  const float* x; long x_ns; int Cbi;
  const float* dy; long dy_ns; int Cbo;
  float* dw;                     // [Cout][Cin][taps]
  int N, Z, S, taps, Cout, Cin;
};

__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
  constexpr int T = 8;                              // spatial tile
  __shared__ float xs[4][T + 2][T + 2][8];          // up to 4 z planes of the input tile (Z <= 4 here)
  __shared__ float ys[4][T][T][8];
  const int tid = threadIdx.x, co = tid & 7, ci = (tid >> 3) & 7, q = tid >> 6;
  const int cob = blockIdx.x, cib = blockIdx.y;
  const int S = a.S, Z = a.Z;
  const long plane = (long)Z * S * S * 8;
  float acc[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) acc[i] = 0.f;
  const int tiles = (S + T - 1) / T;
  for (int n = 0; n < a.N; ++n) {
    const float* xb = a.x + (long)n * a.x_ns + (long)cib * plane;
    const float* yb = a.dy + (long)n * a.dy_ns + (long)cob * plane;
    for (int ty = 0; ty < tiles; ++ty)
      for (int tx = 0; tx < tiles; ++tx) {
        __syncthreads();
        for (int i = tid; i < Z * (T + 2) * (T + 2) * 8; i += 256) {
          const int c8 = i & 7;
          int r = i >> 3;
          const int hx = r % (T + 2); r /= (T + 2);
          const int hy = r % (T + 2);
          const int z = r / (T + 2);
          const int y = ty * T + hy - 1, x = tx * T + hx - 1;
          xs[z][hy][hx][c8] = (y >= 0 && y < S && x >= 0 && x < S) ? xb[((long)(z * S + y) * S + x) * 8 + c8] : 0.f;
        }
        for (int i = tid; i < Z * T * T * 8; i += 256) {
          const int c8 = i & 7;
          int r = i >> 3;
          const int lx = r % T; r /= T;
          const int ly = r % T;
          const int z = r / T;
          const int y = ty * T + ly, x = tx * T + lx;
          ys[z][ly][lx][c8] = (y < S && x < S) ? yb[((long)(z * S + y) * S + x) * 8 + c8] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          const int t = q + 4 * i;
          if (t >= a.taps) break;
          const int kz = a.taps == 1 ? 1 : t / 9, ky = a.taps == 1 ? 1 : (t / 3) % 3, kx = a.taps == 1 ? 1 : t % 3;
          float s_ = 0.f;
          for (int z = 0; z < Z; ++z) {
            const int zi = z + kz - 1;
            if (zi < 0 || zi >= Z) continue;
            for (int ly = 0; ly < T; ++ly)
#pragma unroll
              for (int lx = 0; lx < T; ++lx) s_ = fmaf(ys[z][ly][lx][co], xs[zi][ly + ky][lx + kx][ci], s_);
          }
          acc[i] += s_;
        }
      }
  }
  const int oc = cob * 8 + co, ic = cib * 8 + ci;
  if (oc < a.Cout && ic < a.Cin) {
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int t = q + 4 * i;
      if (t < a.taps) a.dw[((long)oc * a.Cin + ic) * a.taps + t] = acc[i];
    }
  }
}

hipError_t launch_conv_wgrad(const TV& x, const TV& dy, float* dw, int Cin, int Cout, int taps, hipStream_t s) {
  if ((taps != 27 && taps != 1) || x.Z != dy.Z || x.H != dy.H || x.N != dy.N || x.Z > 4 || x.H != x.W) return hipErrorInvalidValue;
  WgradArgs a{x.p, x.nstride, x.Cb, dy.p, dy.nstride, dy.Cb, dw, x.N, x.Z, x.H, taps, Cout, Cin};
  hipLaunchKernelGGL(conv_wgrad_kernel, dim3((unsigned)dy.Cb, (unsigned)x.Cb), dim3(256), 0, s, a);
  return hipGetLastError();
}

// per-channel sum over (n, voxels) of a CB8 tensor: bias gradients.  One workgroup per channel block.
__global__ __launch_bounds__(256) void chan_sum_kernel(const float* x, long x_ns, int N, long vpn, float* out, int C) {
  __shared__ float red[4][8];
  const int cb = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float s[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = 0.f;
  for (long i = tid; i < (long)N * vpn; i += 256) {
    const long n = i / vpn, v = i - n * vpn;
    const float* p = x + n * x_ns + ((long)cb * vpn + v) * 8;
    const f32x4 a0 = *(const f32x4*)p, a1 = *(const f32x4*)(p + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { s[j] += a0[j]; s[4 + j] += a1[j]; }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { s[j] = wave_sum(s[j]); if (lane == 0) red[wv][j] = s[j]; }
  __syncthreads();
  if (tid < 8 && cb * 8 + tid < C) out[cb * 8 + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}
hipError_t launch_chan_sum(const TV& x, float* out, int C, hipStream_t s) {
  hipLaunchKernelGGL(chan_sum_kernel, dim3((unsigned)x.Cb), dim3(256), 0, s, x.p, x.nstride, x.N, (long)x.Z * x.H * x.W, out, C);
  return hipGetLastError();
}

}  // namespace tmk
