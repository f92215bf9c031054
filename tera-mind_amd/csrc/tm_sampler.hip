// Sampler elementwise kernels of the Tera-MIND denoising path (see tm_kernels.hip for the rest).
#include "tm_kernels.h"

namespace tmk {

// ==========================================================================================
// sampler step on interior pixels only (the -1 padded border of the eps re-tiling is cropped
// away by the reference, base.py:389,628).  This translation unit is compiled with
// -ffp-contract=off so that each product and sum rounds exactly like the reference's
// separate torch ops (bit-exact vs the CPU oracle); a `#pragma clang fp contract(off)` is
// not honoured by hipcc under -ffp-contract=fast.
// ==========================================================================================
__global__ __launch_bounds__(256) void sampler_step_kernel(StepCoefs c, const float* xp, const float* eps,
                                                           const float* noise, float* out, int b, int P1, int P2,
                                                           int C, int ps, int mode) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int H = P1 * ps, W = P2 * ps, hp = ps / 2;
  const long tot = (long)b * C * H * W;
  if (i >= tot) return;
  const int X = (int)(i % W);
  long r = i / W;
  const int Y = (int)(r % H); r /= H;
  const int ch = (int)(r % C);
  const int bi = (int)(r / C);
  // padded-grid patch holding this pixel
  const int Yp = Y + hp, Xp = X + hp;
  const int pi = Yp / ps, pj = Xp / ps;
  const long xo = ((((long)bi * (P1 + 1) + pi) * (P2 + 1) + pj) * C + ch) * ps * ps + (long)(Yp - pi * ps) * ps + (Xp - pj * ps);
  const int ei = Y / ps, ej = X / ps;
  const long eo = ((((long)bi * P1 + ei) * P2 + ej) * C + ch) * ps * ps + (long)(Y - ei * ps) * ps + (X - ej * ps);
  const float xv = xp[xo], ev = eps[eo];
  float x0 = c.c_recip * xv - c.c_recipm1 * ev;
  x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
  float o;
  if (mode == 0) {
    const float mean = c.pm1 * x0 + c.pm2 * xv;
    o = mean;
    if (noise) o = mean + c.sigma * noise[xo];
  } else {
    const float e2 = (c.c_recip * xv - x0) / c.c_recipm1;
    o = x0 * c.sab_prev + c.s1m_ab_prev * e2;
  }
  out[i] = o;
}
hipError_t launch_sampler_step(const StepCoefs& c, const float* x_patches, const float* eps, const float* noise,
                               float* out, int b, int P1, int P2, int C, int ps, int mode, hipStream_t s) {
  const long tot = (long)b * C * P1 * ps * P2 * ps;
  hipLaunchKernelGGL(sampler_step_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, c, x_patches, eps,
                     noise, out, b, P1, P2, C, ps, mode);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void pad_patchify_kernel(const float* img, float* pt, int b, int C, int P1, int P2,
                                                           int ps, float pad) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long tot = (long)b * (P1 + 1) * (P2 + 1) * C * ps * ps;
  if (i >= tot) return;
  const int x = (int)(i % ps);
  long r = i / ps;
  const int y = (int)(r % ps); r /= ps;
  const int ch = (int)(r % C); r /= C;
  const int pj = (int)(r % (P2 + 1)); r /= (P2 + 1);
  const int pi = (int)(r % (P1 + 1));
  const int bi = (int)(r / (P1 + 1));
  const int hp = ps / 2;
  const int Y = pi * ps + y - hp, X = pj * ps + x - hp;
  float v = pad;
  if (Y >= 0 && Y < P1 * ps && X >= 0 && X < P2 * ps) v = img[(((long)bi * C + ch) * P1 * ps + Y) * P2 * ps + X];
  pt[i] = v;
}
hipError_t launch_pad_patchify(const float* img, float* patches, int b, int C, int P1, int P2, int ps, float pad,
                               hipStream_t s) {
  const long tot = (long)b * (P1 + 1) * (P2 + 1) * C * ps * ps;
  hipLaunchKernelGGL(pad_patchify_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, img, patches, b, C,
                     P1, P2, ps, pad);
  return hipGetLastError();
}

}  // namespace tmk
