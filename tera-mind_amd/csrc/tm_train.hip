// Training slice: backward kernels of one ResBlock (model/MBAblocks.py:237-299,302-368) --
//   prep_bwd_kernel    backward of  y = Dropout(SiLU(RMSNorm_C(x) * w * (1 + scale) + shift))  (in_layers[0:2], out_layers[0:3])
//   conv_wgrad_kernel  dW of Conv3d(k = 3x3x3 pad 1, Z <= 4) and of the 1x1x1 skip conv
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


// ==================================================================================================================
// AttnBlock training slice (model/MBAblocks.py:428-514,517-601,608-614): elementwise pieces, the per-voxel adaLN
// modulate(norm(x)) backward, and the windowed cross-attention core forward + backward.  fp32, functional (not tuned):
// every Linear of the block is a 1x1x1 conv and runs, forward and backward, on the conv kernels above.
// ==================================================================================================================

// ---- elementwise ops on flat fp32 buffers (CB8 tensors of one geometry) ----
//   0 GATE_ADD  o1 = a + b * c          x + gate * value                         (MBAblocks.py:488-489)
//   1 MUL2      o1 = a * b, o2 = a * c  d(value) = d * gate, d(gate) = d * value
//   2 GELU      o1 = gelu_tanh(a)       timm Mlp act (approx_gelu, MBAblocks.py:18)
//   3 GELU_BWD  o1 = a * gelu_tanh'(b)
//   4 SILU      o1 = silu(a)            adaLN_modulation[0] (MBAblocks.py:463)
//   5 SILU_BWD  o1 = a * silu'(b)
//   6 ADD       o1 = a + b
//   7 MUL4      o1 = 4 a                adjoint of AvgPool(1,2,2) composed from the x2 nearest upsample (and vice versa:
//   8 DIV4      o1 = a / 4              up2^T = 4 avgpool, avgpool^T = up2 / 4)
__global__ __launch_bounds__(256) void ew_kernel(int op, const float* a, const float* b, const float* c, float* o1, float* o2, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float av = a[i];
    switch (op) {
      case 0: o1[i] = av + b[i] * c[i]; break;
      case 1: o1[i] = av * b[i]; o2[i] = av * c[i]; break;
      case 2: o1[i] = gelu_tanh_f(av); break;
      case 3: {
        const float x = b[i], kB = 0.7978845608028654f, kK = 0.044715f;
        const float u = kB * (x + kK * x * x * x), th = tanhf(u);
        o1[i] = av * (0.5f * (1.0f + th) + 0.5f * x * (1.0f - th * th) * kB * (1.0f + 3.0f * kK * x * x));
        break;
      }
      case 4: o1[i] = silu_f(av); break;
      case 5: { const float x = b[i], sg = 1.0f / (1.0f + expf(-x)); o1[i] = av * sg * (1.0f + x * (1.0f - sg)); break; }
      case 6: o1[i] = av + b[i]; break;
      case 7: o1[i] = 4.0f * av; break;
      default: o1[i] = 0.25f * av; break;
    }
  }
}
hipError_t launch_ew(int op, const float* a, const float* b, const float* c, float* o1, float* o2, long n, hipStream_t s) {
  if (op < 0 || op > 8 || !a || !o1 || n < 0) return hipErrorInvalidValue;
  long g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(ew_kernel, dim3((unsigned)g), dim3(256), 0, s, op, a, b, c, o1, o2, n);
  return hipGetLastError();
}

// ---- modulate(norm, x, shift, scale) backward with PER-VOXEL shift / scale (MBAblocks.py:608-614) ----
//   forward  y = RMSNorm_C(x) * w * (1 + scale) + shift         scale, shift: CB8 tensors of x's geometry
//   backward dscale = g * n (n = xh * w), dshift = g, dn = g * (1 + scale), dw[c] += sum_v dn * xh,
//            dx = rstd * (dn * w - xh * mean_c(dn * w * xh))
// Same structure as prep_bwd_kernel: lane = voxel, four waves split the channel blocks; dw through the two-stage reduction.
struct ModNormBwdArgs {
  const float* x; const float* g; const float* w; const float* scale;
  float* dx; float* dscale; float* dshift; float* part_dw;
  long ns; int N, Cb, Z, S; float inv_c;
};
__global__ __launch_bounds__(256) void modnorm_bwd_kernel(ModNormBwdArgs a) {
  __shared__ float red[4][64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const long vpn = (long)a.Z * a.S * a.S;
  const long vidx = (long)blockIdx.x * 64 + lane;
  const bool valid = vidx < vpn * a.N;
  const int n = valid ? (int)(vidx / vpn) : 0;
  const long off = valid ? (vidx - (long)n * vpn) * 8 : 0;
  const long plane = vpn * 8;
  float ssq = 0.f;
  if (valid)
    for (int cb = wv; cb < a.Cb; cb += 4) {
      const float* p = a.x + (long)n * a.ns + (long)cb * plane + off;
#pragma unroll
      for (int j = 0; j < 8; ++j) ssq += p[j] * p[j];
    }
  red[wv][lane] = ssq;
  __syncthreads();
  const float rstd = 1.0f / sqrtf((red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]) * a.inv_c + TM_EPS);
  __syncthreads();
  float dot = 0.f;
  for (int cb = wv; cb < a.Cb; cb += 4) {
    const long o = (long)n * a.ns + (long)cb * plane + off;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = cb * 8 + j;
      float xh = 0.f, dn = 0.f;
      if (valid) { xh = a.x[o + j] * rstd; dn = a.g[o + j] * (1.0f + a.scale[o + j]); }
      dot += dn * a.w[c] * xh;
      const float dwv = wave_sum(dn * xh);
      if (lane == 0) a.part_dw[(long)blockIdx.x * a.Cb * 8 + c] = dwv;
    }
  }
  red[wv][lane] = dot;
  __syncthreads();
  const float mean_dot = (red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]) * a.inv_c;
  if (!valid) return;
  for (int cb = wv; cb < a.Cb; cb += 4) {
    const long o = (long)n * a.ns + (long)cb * plane + off;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = cb * 8 + j;
      const float xh = a.x[o + j] * rstd, gv = a.g[o + j];
      a.dscale[o + j] = gv * xh * a.w[c];
      a.dshift[o + j] = gv;
      a.dx[o + j] = rstd * (gv * (1.0f + a.scale[o + j]) * a.w[c] - xh * mean_dot);
    }
  }
}
hipError_t launch_modnorm_bwd(const TV& x, const float* g, const float* w, const float* scale, float* dx, float* dscale, float* dshift,
                              float* dw, int C_real, float* scratch, hipStream_t s) {
  const long vox = (long)x.N * x.Z * x.H * x.W, nwg = (vox + 63) / 64;
  ModNormBwdArgs a{x.p, g, w, scale, dx, dscale, dshift, scratch, x.nstride, x.N, x.Cb, x.Z, x.H, 1.0f / (float)C_real};
  hipLaunchKernelGGL(modnorm_bwd_kernel, dim3((unsigned)nwg), dim3(256), 0, s, a);
  hipLaunchKernelGGL(prep_bwd_reduce_dw_kernel, dim3((unsigned)((x.Cb * 8 + 63) / 64)), dim3(64), 0, s, scratch, nwg, x.Cb * 8, dw);
  return hipGetLastError();
}

// ---- windowed cross-attention core (MBAblocks.py:551-601, n_h = 2, one head): forward and backward ----
//   qh = RMSNorm_C(q) * qw, kh = RMSNorm_C(k) * kw;  S = qh kh^T / C;  P = softmax_j S;  o = P v        per window of
//   T = Z (S/2)^2 tokens (2 x 2 windows over (h, w), all z).  One workgroup per (patch, window), T <= 128, C <= 512.
//   Backward recomputes P:  dv = P^T do;  dP = do v^T;  dS = P (dP - rowsum(dP P)) / C;  dqh = dS kh;  dkh = dS^T qh;
//   RMSNorm backward per token (g = dqh * qw: dq = r g - q r^3 mean_c(g q));  d(qw)[c] = sum_tokens dqh q r -- per-workgroup
//   partials, reduced in index order by prep_bwd_reduce_dw_kernel.
struct AttnTrainArgs {
  const float *q, *k, *v, *qw, *kw, *dout;
  float *o;                            // forward output (BWD == false)
  float *dq, *dk, *dv, *part_qw, *part_kw;
  long ns; int C, Cb, Z, S, T;
};
constexpr int AT_CH = 16;              // channels per staged chunk
template <bool BWD>
__global__ __launch_bounds__(256) void attn_train_kernel(AttnTrainArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int T = a.T, C = a.C, tid = threadIdx.x;
  float* Pm = sm;                                  // [T][T]
  float* Dm = Pm + (BWD ? T * T : 0);              // [T][T] (backward)
  float* A = Dm + T * T;                           // [T][AT_CH] staged operand (row scaled)
  float* B = A + T * AT_CH;                        // [T][AT_CH]
  float* rq = B + T * AT_CH;                       // [T]
  float* rk = rq + T;
  float* rowdot = rk + T;                          // [T]
  float* pdot = rowdot + T;                        // [T][8] partial dots
  int* tokoff = (int*)(pdot + T * 8);              // [T]
  const int S = a.S, hs = S / 2;
  const int n = blockIdx.x >> 2, win = blockIdx.x & 3, wy = win >> 1, wx = win & 1;
  const long plane = (long)a.Z * S * S * 8;
  const long nb = (long)n * a.ns;
  if (tid < T) {
    const int z = tid / (hs * hs), r = tid - z * hs * hs, yl = r / hs, xl = r - yl * hs;
    tokoff[tid] = ((z * S + wy * hs + yl) * S + wx * hs + xl) * 8;
  }
  __syncthreads();
  auto at = [&](const float* base, int t, int c) -> float { return base[nb + (long)(c >> 3) * plane + tokoff[t] + (c & 7)]; };
  auto put = [&](float* base, int t, int c, float v) { base[nb + (long)(c >> 3) * plane + tokoff[t] + (c & 7)] = v; };
  for (int t = tid; t < 2 * T; t += 256) {
    const float* src = t < T ? a.q : a.k;
    const int tt = t < T ? t : t - T;
    float ss = 0.f;
    for (int c = 0; c < C; ++c) { const float v = at(src, tt, c); ss += v * v; }
    (t < T ? rq : rk)[tt] = 1.0f / sqrtf(ss / (float)C + TM_EPS);
  }
  __syncthreads();
  // stage rows [T][AT_CH] of `src` for channels c0 .. c0 + AT_CH: value * rowscale[t] * colscale[c] (either may be null)
  auto stage = [&](float* dst, const float* src, int c0, const float* rs, const float* cs) {
    for (int e = tid; e < T * AT_CH; e += 256) {
      const int t = e / AT_CH, cc = e - t * AT_CH, c = c0 + cc;
      float v = 0.f;
      if (c < C) { v = at(src, t, c); if (rs) v *= rs[t]; if (cs) v *= cs[c]; }
      dst[e] = v;
    }
  };
  // M[i][j] = sum_c X[i][c] Y[j][c] over all channels (X, Y staged chunk by chunk); thread (ti, tj) owns a TT x TT tile
  const int TT = T / 16, ti = tid >> 4, tj = tid & 15;
  auto gemm_nt = [&](float* M, const float* X, const float* xr, const float* xc, const float* Y, const float* yr, const float* yc, float mul) {
    float acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
    for (int c0 = 0; c0 < C; c0 += AT_CH) {
      __syncthreads();
      stage(A, X, c0, xr, xc);
      stage(B, Y, c0, yr, yc);
      __syncthreads();
      for (int cc = 0; cc < AT_CH; ++cc)
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (i < TT) {
            const float xv = A[(ti * TT + i) * AT_CH + cc];
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if (j < TT) acc[i][j] = fmaf(xv, B[(tj * TT + j) * AT_CH + cc], acc[i][j]);
          }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (i < TT && j < TT) M[(ti * TT + i) * T + tj * TT + j] = acc[i][j] * mul;
    __syncthreads();
  };
  // P = softmax(qh kh^T / C)
  gemm_nt(Pm, a.q, rq, a.qw, a.k, rk, a.kw, 1.0f / (float)C);
  if (tid < T) {
    float m = -INFINITY;
    for (int j = 0; j < T; ++j) m = fmaxf(m, Pm[tid * T + j]);
    float sum = 0.f;
    for (int j = 0; j < T; ++j) { const float e = expf(Pm[tid * T + j] - m); Pm[tid * T + j] = e; sum += e; }
    const float inv = 1.0f / sum;
    for (int j = 0; j < T; ++j) Pm[tid * T + j] *= inv;
  }
  __syncthreads();
  // out[r][c] = sum_k M(r, k) * Y[k][c] for every channel, M(r, k) = trans ? Mat[k][r] : Mat[r][k]; thread -> (row, channel subset)
  const int NTR = 256 / T, row = tid % T, sub = tid / T, cps = AT_CH / NTR;       // NTR in {2, 4, 8}: cps in {8, 4, 2}
  auto matmul_out = [&](const float* Mat, bool trans, const float* Y, const float* yr, const float* yc, float* out, const float* rdot_src,
                        const float* rdot_w, float* dotacc) {
    for (int c0 = 0; c0 < C; c0 += AT_CH) {
      __syncthreads();
      stage(B, Y, c0, yr, yc);
      __syncthreads();
      for (int u = 0; u < cps; ++u) {
        const int cc = sub * cps + u, c = c0 + cc;
        if (c >= C) continue;
        float s_ = 0.f;
        for (int k = 0; k < T; ++k) s_ = fmaf(trans ? Mat[k * T + row] : Mat[row * T + k], B[k * AT_CH + cc], s_);
        put(out, row, c, s_);
        if (dotacc) *dotacc += s_ * rdot_w[c] * at(rdot_src, row, c);
      }
    }
    __syncthreads();
  };
  if (!BWD) {
    matmul_out(Pm, false, a.v, nullptr, nullptr, a.o, nullptr, nullptr, nullptr);
    return;
  }
  // dv = P^T dout
  matmul_out(Pm, true, a.dout, nullptr, nullptr, a.dv, nullptr, nullptr, nullptr);
  // dP = dout v^T ; dS = P (dP - rowsum(dP P)) / C
  gemm_nt(Dm, a.dout, nullptr, nullptr, a.v, nullptr, nullptr, 1.0f);
  if (tid < T) {
    float rd = 0.f;
    for (int j = 0; j < T; ++j) rd += Dm[tid * T + j] * Pm[tid * T + j];
    rowdot[tid] = rd;
  }
  __syncthreads();
  for (int e = tid; e < T * T; e += 256) { const int i = e / T; Dm[e] = Pm[e] * (Dm[e] - rowdot[i]) / (float)C; }
  __syncthreads();
  // dqh = dS kh (stored in dq for now), partial dots of g q with g = dqh * qw; then the RMSNorm backward in place
  for (int pass = 0; pass < 2; ++pass) {
    const float* Xsrc = pass == 0 ? a.q : a.k;           // the tensor being differentiated
    const float* Ysrc = pass == 0 ? a.k : a.q;           // the other operand (normalised)
    const float* xr = pass == 0 ? rq : rk; const float* yr = pass == 0 ? rk : rq;
    const float* xw = pass == 0 ? a.qw : a.kw; const float* yw = pass == 0 ? a.kw : a.qw;
    float* dX = pass == 0 ? a.dq : a.dk;
    float* part = pass == 0 ? a.part_qw : a.part_kw;
    float mydot = 0.f;
    matmul_out(Dm, pass == 1, Ysrc, yr, yw, dX, Xsrc, xw, &mydot);
    pdot[row * 8 + sub] = mydot;
    __syncthreads();
    if (tid < T) { float d = 0.f; for (int u = 0; u < NTR; ++u) d += pdot[tid * 8 + u]; rowdot[tid] = d / (float)C; }
    __syncthreads();
    // per-channel partial of d(norm weight): sum over the window's tokens of dXh[t][c] * x[t][c] * r[t]   (dXh still in dX)
    for (int c = tid; c < a.Cb * 8; c += 256) {
      float s_ = 0.f;
      if (c < C) for (int t = 0; t < T; ++t) s_ += at(dX, t, c) * at(Xsrc, t, c) * xr[t];
      part[(long)blockIdx.x * a.Cb * 8 + c] = s_;
    }
    __syncthreads();
    // dx = r g - x r^3 mean_c(g x),  g = dXh * w
    for (int e = tid; e < T * C; e += 256) {
      const int t = e / C, c = e - t * C;
      const float r = xr[t], xv = at(Xsrc, t, c);
      put(dX, t, c, r * at(dX, t, c) * xw[c] - xv * r * r * r * rowdot[t]);
    }
    __syncthreads();
  }
}
size_t attn_train_lds_bytes(int T, bool bwd) { return (size_t)((bwd ? 2 : 1) * T * T + 2 * T * AT_CH + 3 * T + T * 8 + T) * 4; }
hipError_t launch_attn_train(const TV& q, const TV& k, const TV& v, const float* qw, const float* kw, const float* dout, float* o,
                             float* dq, float* dk, float* dv, float* dqw, float* dkw, float* scratch, bool bwd, hipStream_t s) {
  const int S = q.H, T = q.Z * (S / 2) * (S / 2);
  if (q.H != q.W || (S & 1) || (T != 32 && T != 64 && T != 128) || q.C > 512 || k.nstride != q.nstride || v.nstride != q.nstride)
    return hipErrorInvalidValue;
  const long nwg = (long)q.N * 4;
  AttnTrainArgs a{q.p, k.p, v.p, qw, kw, dout, o, dq, dk, dv, scratch, scratch ? scratch + nwg * q.Cb * 8 : nullptr, q.nstride, q.C, q.Cb,
                  q.Z, S, T};
  const size_t lds = attn_train_lds_bytes(T, bwd);
  static DevOnce attr_done;
  if (attr_done.need()) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_train_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_train_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_done.mark();
  }
  if (bwd) {
    hipLaunchKernelGGL(attn_train_kernel<true>, dim3((unsigned)nwg), dim3(256), lds, s, a);
    hipLaunchKernelGGL(prep_bwd_reduce_dw_kernel, dim3((unsigned)((q.Cb * 8 + 63) / 64)), dim3(64), 0, s, a.part_qw, nwg, q.Cb * 8, dqw);
    hipLaunchKernelGGL(prep_bwd_reduce_dw_kernel, dim3((unsigned)((q.Cb * 8 + 63) / 64)), dim3(64), 0, s, a.part_kw, nwg, q.Cb * 8, dkw);
  } else {
    hipLaunchKernelGGL(attn_train_kernel<false>, dim3((unsigned)nwg), dim3(256), lds, s, a);
  }
  return hipGetLastError();
}


// ==================================================================================================================
// Small dense pieces of the training path that are not convs over the patch volume: the Linears over [tokens][features]
// rows (time embedding, ResBlock.emb_layers, the gene-gene AttnBlock of model/unet_ours.py:277-323) and that block's
// row-wise RMSNorm / softmax.  fp32 VALU, deterministic (no split-K, fixed-order reductions).
// ==================================================================================================================

// C[b](m, n) = alpha * sum_k A[b](m, k) * B[b](k, n) (+ bias) (+ C): every operand through element strides, so the same kernel
// serves y = x W^T, dx = dy W, dW = dy^T x, q q^T, P v and their transposes.  64 x 64 tile per workgroup, 4 x 4 per thread.
struct GemmArgs {
  const float *A, *B, *bias; float* C;
  int M, N, K;
  long sam, sak, sbk, sbn, scm, scn, sab, sbb, scb;
  int bias_mode;      // 0 none, 1 bias[n], 2 bias[m]
  int accumulate;     // C += ...
  float alpha;
};
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs a) {
  __shared__ float As[16][65], Bs[16][65];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const float* A = a.A + (long)blockIdx.z * a.sab;
  const float* B = a.B + (long)blockIdx.z * a.sbb;
  float* Cp = a.C + (long)blockIdx.z * a.scb;
  float acc[4][4] = {};
  for (int k0 = 0; k0 < a.K; k0 += 16) {
    for (int e = tid; e < 16 * 64; e += 256) {
      const int kk = e >> 6, r = e & 63;
      const int k = k0 + kk;
      As[kk][r] = (k < a.K && m0 + r < a.M) ? A[(long)(m0 + r) * a.sam + (long)k * a.sak] : 0.f;
      Bs[kk][r] = (k < a.K && n0 + r < a.N) ? B[(long)k * a.sbk + (long)(n0 + r) * a.sbn] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { av[i] = As[kk][ty * 4 + i]; bv[i] = Bs[kk][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + ty * 4 + i, n = n0 + tx * 4 + j;
      if (m >= a.M || n >= a.N) continue;
      float v = acc[i][j] * a.alpha;
      if (a.bias_mode == 1) v += a.bias[n];
      else if (a.bias_mode == 2) v += a.bias[m];
      float* c = Cp + (long)m * a.scm + (long)n * a.scn;
      *c = a.accumulate ? *c + v : v;
    }
}
hipError_t launch_gemm_f32(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, const long* st, int batch,
                           int bias_mode, int accumulate, float alpha, hipStream_t s) {
  if (!A || !B || !C || M < 1 || N < 1 || K < 1 || batch < 1 || (bias_mode && !bias)) return hipErrorInvalidValue;
  GemmArgs a{A, B, bias, C, M, N, K, st[0], st[1], st[2], st[3], st[4], st[5], st[6], st[7], st[8], bias_mode, accumulate, alpha};
  hipLaunchKernelGGL(gemm_f32_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64), (unsigned)batch), dim3(256), 0, s, a);
  return hipGetLastError();
}

// Row-wise ops on [rows][D] fp32 (one wave per row, four rows per workgroup):
//   0 RMSNORM      y = x * rsqrt(mean_d x^2 + eps) * w                                  (LlamaRMSNorm dim=-1, MBAblocks.py:35-43)
//   1 RMSNORM_BWD  dx = r (g w - xh mean_d(g w xh)),  part[wg][d] = sum over the workgroup's rows of g * xh
//   2 SOFTMAX      y = softmax_d(x)
//   3 SOFTMAX_BWD  dx = y_saved * (g - sum_d g y_saved)      (x = the saved probabilities)
__global__ __launch_bounds__(256) void rows_kernel(int op, const float* x, const float* w, const float* g, float* y, float* part, long rows,
                                                   int D) {
  extern __shared__ float sh[];                      // op 1: [4][D] per-row products
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long r = (long)blockIdx.x * 4 + wv;
  const bool live = r < rows;
  const float* xr = x + (live ? r : 0) * D;
  if (op == 0 || op == 1) {
    float ss = 0.f;
    if (live) for (int d = lane; d < D; d += 64) ss += xr[d] * xr[d];
    ss = wave_sum(ss);
    const float rstd = 1.0f / sqrtf(ss / (float)D + TM_EPS);
    if (op == 0) {
      if (live) for (int d = lane; d < D; d += 64) y[r * D + d] = xr[d] * rstd * w[d];
      return;
    }
    float dot = 0.f;
    if (live) for (int d = lane; d < D; d += 64) dot += g[r * D + d] * w[d] * xr[d] * rstd;
    dot = wave_sum(dot) / (float)D;
    for (int d = lane; d < D; d += 64) {
      float pv = 0.f;
      if (live) {
        const float xh = xr[d] * rstd, gv = g[r * D + d];
        y[r * D + d] = rstd * (gv * w[d] - xh * dot);
        pv = gv * xh;
      }
      sh[wv * D + d] = pv;
    }
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += 256) part[(long)blockIdx.x * D + d] = (sh[d] + sh[D + d]) + (sh[2 * D + d] + sh[3 * D + d]);
    return;
  }
  if (!live) return;
  if (op == 2) {
    float m = -INFINITY;
    for (int d = lane; d < D; d += 64) m = fmaxf(m, xr[d]);
    for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float sum = 0.f;
    for (int d = lane; d < D; d += 64) sum += expf(xr[d] - m);
    sum = wave_sum(sum);
    for (int d = lane; d < D; d += 64) y[r * D + d] = expf(xr[d] - m) / sum;
  } else {
    float dot = 0.f;
    for (int d = lane; d < D; d += 64) dot += g[r * D + d] * xr[d];
    dot = wave_sum(dot);
    for (int d = lane; d < D; d += 64) y[r * D + d] = xr[d] * (g[r * D + d] - dot);
  }
}
hipError_t launch_rows(int op, const float* x, const float* w, const float* g, float* y, float* dw, float* scratch, long rows, int D,
                       hipStream_t s) {
  if (op < 0 || op > 3 || !x || !y || rows < 1 || D < 1 || D > 8192) return hipErrorInvalidValue;
  if ((op <= 1 && !w) || ((op == 1 || op == 3) && !g) || (op == 1 && (!dw || !scratch))) return hipErrorInvalidValue;
  const long nwg = (rows + 3) / 4;
  hipLaunchKernelGGL(rows_kernel, dim3((unsigned)nwg), dim3(256), op == 1 ? (size_t)4 * D * sizeof(float) : 0, s, op, x, w, g, y, scratch, rows, D);
  if (op == 1) hipLaunchKernelGGL(prep_bwd_reduce_dw_kernel, dim3((unsigned)((D + 63) / 64)), dim3(64), 0, s, scratch, nwg, D, dw);
  return hipGetLastError();
}


// ==================================================================================================================
// Optimizer step of the reference's training loop (experiment.py:207-219, 394-414): torch.nn.utils.clip_grad_norm_ over all
// parameters, then torch.optim.Adam(lr, weight_decay) -- on one flat fp32 arena of parameters / gradients / moments.
// ==================================================================================================================

// part[wg] = sum of squares of this workgroup's grid-stride share, then one thread adds the partials in index order
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* x, long n, float* part) {
  __shared__ float red[4];
  float s_ = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s_ = fmaf(x[i], x[i], s_);
  s_ = wave_sum(s_);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s_;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
hipError_t launch_sumsq(const float* x, long n, float* out, float* scratch, int nwg, hipStream_t s) {
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3((unsigned)nwg), dim3(256), 0, s, x, n, scratch);
  hipLaunchKernelGGL(prep_bwd_reduce_dw_kernel, dim3(1), dim3(64), 0, s, scratch, (long)nwg, 1, out);
  return hipGetLastError();
}

// torch.optim.Adam._single_tensor_adam (amsgrad off, maximize off):  g' = g * gscale + wd * p;  m = m + (g' - m)(1 - b1);
// v = b2 v + (1 - b2) g'^2;  p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                                                   float wd, float bc1, float bc2_sqrt, float gscale) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float pv = p[i];
    const float gv = fmaf(wd, pv, g[i] * gscale);
    const float mv = m[i] + (gv - m[i]) * (1.0f - b1);
    const float vv = b2 * v[i] + (1.0f - b2) * gv * gv;
    m[i] = mv; v[i] = vv;
    p[i] = pv - (lr / bc1) * (mv / (sqrtf(vv) / bc2_sqrt + eps));
  }
}
hipError_t launch_adam(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, float wd, int step,
                       float gscale, hipStream_t s) {
  if (!p || !g || !m || !v || n < 1 || step < 1) return hipErrorInvalidValue;
  const float bc1 = 1.0f - powf(b1, (float)step), bc2 = 1.0f - powf(b2, (float)step);
  long grid = (n + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)grid), dim3(256), 0, s, p, g, m, v, n, lr, b1, b2, eps, wd, bc1, sqrtf(bc2), gscale);
  return hipGetLastError();
}

}  // namespace tmk
