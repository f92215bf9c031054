// Attention kernels of the Tera-MIND denoising hot path (fp32; the bf16 attention core lives in tm_conv_bf16.hip):
//   gene_attn_mfma_kernel        gene-gene attention block, fused, D = 64 (the checkpoint geometry)
//   gene_attn_generic_kernel     the same block for any G <= 512, D <= 512
//   window_attn_mfma_kernel      windowed gene-patch cross attention core, T = 128 (fp32 MFMA)
//   window_attn_kernel<32>       T = 32 (middle block)
//   window_attn_generic_kernel   any T <= 512
//   rna_mid_kernel               rna_h[:, :, 1:-1] of the attention-map model
#include "tm_device.h"

#include <stdlib.h>
#include <string.h>

namespace tmk {

// ==========================================================================================
// Gene-gene attention block (AttnBlock gene_trans=False: model/MBAblocks.py:492-501 and
// Attention.forward :551-601 with k = q, q_norm on both, scale 1/64, no residuals), one
// workgroup per patch, tokens = G genes x D = 64 features (z h w); the result is written as the
// CB8 tensor [B][ceil(G/8)][zs][gn][gn][8] that down_z consumes.  The softmax row lives
// across the 64 lanes of a wave (4 keys per lane) and is reduced with wave shuffles.
// P.V is computed as (P.tok).Wv^T + bv (rows of P sum to 1), so V is never materialised.
// ==========================================================================================
struct GeneArgs {
  const float* rna; int B, gn, zs, G;
  GeneW w;
  float* out_tok; float* attn_map; float* scratch;
  int zlo, zhi;
};
#define GENE_D 64

// gridDim.y workgroups per patch share the rows of passes B/C (2 when the batch alone would leave CUs idle)
// token (n, gene g, feature d = (z h w)) inside the CB8 tensor [B][ceil(G/8)][zs][gn][gn][8]
__device__ __forceinline__ long gene_tok_idx(int n, int g, int d, int Gb) {
  return (((long)n * Gb + (g >> 3)) * GENE_D + d) * 8 + (g & 7);
}

// ---- MFMA form of the gene-gene attention block --------------------------------------------------------
// Everything is computed TRANSPOSED (features / keys on the rows, the 32 genes of a query block on the
// columns), so that each 32x32 result tile -- column = lane, rows = accumulator registers -- is directly
// the B operand of the next product, which always sums over the previous result's ROW index:
//   q^T = Wq.tok^T   ->  S^T = qn.qn^T (keys x queries)  ->  softmax over rows (in-lane + one xor-32 shuffle)
//   PT^T = tok^T.P^T ->  ov^T = Wv.PT^T  ->  op^T = Wp.ov^T  ->  norm2  ->  y1^T = W1.h^T  ->  y2^T = W2.gelu(y1^T)
// The A operands are weight / token fragments: lanes = rows, contiguous in the pre-transposed weight
// matrices ([in][out]) and in the token image.  k pairs are (row r of the lower half, row r + 4 of the upper
// half) of an accumulator register -- any pairing is valid as long as A uses the same one.
// One workgroup per patch (x gridDim.y query-block shares), 4 waves, wave w owns query blocks w, w+4.
#define GTP 68                                           // LDS row pitch (floats) of the token / q images
#define GROWS 232                                        // rows kept in LDS (genes padded to a multiple of 8)
__device__ __forceinline__ int mfma_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__global__ __launch_bounds__(256) void gene_attn_mfma_kernel(GeneArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* tok = sm;                       // [GROWS][GTP]
  float* qn = tok + GROWS * GTP;         // [GROWS][GTP]
  float* cst = qn + GROWS * GTP;         // bq 64 | qnorm 64 | bv 64 | bp 64 | norm2 64 | b2 64 | b1 256
  const int G = a.G, Gb = (G + 7) / 8;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int i32 = lane & 31, h = lane >> 5;
  const int n = blockIdx.x;
  const int gg = a.gn * a.gn;
  const long rbase = (long)n * gg * a.zs * 500;
  for (int i = tid; i < GROWS * GENE_D; i += 256) {
    const int d = i / GROWS, g = i - d * GROWS;
    const int z = d / gg, hw = d - z * gg;
    float v = 0.f;
    if (g < G && z >= a.zlo && z < a.zhi) v = a.rna[rbase + ((long)hw * a.zs + z) * 500 + g];
    tok[g * GTP + d] = v;
  }
  if (tid < 64) {
    cst[tid] = a.w.bq[tid]; cst[64 + tid] = a.w.qnorm[tid];
    if (a.out_tok) {
      cst[128 + tid] = a.w.bv[tid]; cst[192 + tid] = a.w.bp[tid]; cst[256 + tid] = a.w.norm2[tid]; cst[320 + tid] = a.w.b2[tid];
    }
  }
  if (a.out_tok) cst[384 + tid] = a.w.b1[tid];
  __syncthreads();
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // ---- (1) q^T[j][g] = Wq[j][:] . tok[g][:] + bq[j];  qn = RMSNorm over j (rows) * w ----
  for (int gt = wv; gt < 8; gt += 4) {
    const int g = gt * 32 + i32;
    // the weight fragments are loop invariant: without laundering the pointer LICM hoists all of them out of the
    // loop (over a thousand live registers for the whole kernel) and spills
    const float* wq_t = a.w.wq_t;
    asm volatile("" : "+s"(wq_t));
    f32x16 qa[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int r = 0; r < 16; ++r) qa[jt][r] = 0.f;
#pragma unroll
    for (int k0 = 0; k0 < GENE_D; k0 += 8) {
      const f32x4 bf = (g < GROWS) ? *(const f32x4*)(tok + g * GTP + k0 + 4 * h) : zero4;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const float* wrow = wq_t + (k0 + 4 * h + kk) * GENE_D + i32;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) qa[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wrow[jt * 32], bf[kk], qa[jt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    float ss = 0.f;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int r = 0; r < 16; ++r) { qa[jt][r] += cst[jt * 32 + mfma_row(r, h)]; ss += qa[jt][r] * qa[jt][r]; }
    ss += __shfl_xor(ss, 32, 64);
    const float rstd = 1.0f / sqrtf(ss * (1.0f / GENE_D) + TM_EPS);
    if (g < GROWS) {
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const int j0 = jt * 32 + 8 * q4 + 4 * h;
          f32x4 o;
#pragma unroll
          for (int c = 0; c < 4; ++c) o[c] = cst[64 + j0 + c] * (qa[jt][4 * q4 + c] * rstd);
          *(f32x4*)(qn + g * GTP + j0) = o;
        }
    }
  }
  __syncthreads();

  // ---- per query block of 32 genes ----
  for (int qb = wv + 4 * blockIdx.y; qb < 8; qb += 4 * gridDim.y) {
    const int g = qb * 32 + i32;                          // this lane's gene (column)
    const bool gok = g < G;
    const float *wv_t = a.w.wv_t, *wp_t = a.w.wp_t, *w1_t = a.w.w1_t, *w2_t = a.w.w2_t;
    asm volatile("" : "+s"(wv_t), "+s"(wp_t), "+s"(w1_t), "+s"(w2_t));
    // (2) S^T[u][g] = qn[u].qn[g] / 64
    f32x16 sacc[8];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[kt][r] = 0.f;
#pragma unroll
    for (int d0 = 0; d0 < GENE_D; d0 += 8) {
      const f32x4 bf = (g < GROWS) ? *(const f32x4*)(qn + g * GTP + d0 + 4 * h) : zero4;
#pragma unroll
      for (int kt = 0; kt < 8; ++kt) {
        const int u = kt * 32 + i32;
        const f32x4 af = (u < GROWS) ? *(const f32x4*)(qn + u * GTP + d0 + 4 * h) : zero4;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk], bf[kk], sacc[kt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);         // bound load hoisting: the unrolled body would otherwise spill
    }
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int u = kt * 32 + mfma_row(r, h);
        sacc[kt][r] = (u < G) ? sacc[kt][r] * 0.015625f : -INFINITY;      // (q*scale).k*scale, scale = 1/8
        m = fmaxf(m, sacc[kt][r]);
      }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float ssum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) { sacc[kt][r] = expf(sacc[kt][r] - m); ssum += sacc[kt][r]; }
    __builtin_amdgcn_sched_barrier(0);
    ssum += __shfl_xor(ssum, 32, 64);
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[kt][r] = sacc[kt][r] / ssum;
    if (a.attn_map && gok) {
      float* mp = a.attn_map + ((long)n * G + g) * G;
#pragma unroll
      for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int u = kt * 32 + mfma_row(r, h);
          if (u < G) mp[u] = sacc[kt][r];
        }
    }
    if (!a.out_tok) continue;
    // (3) PT^T[d][g] = sum_u tok[u][d] P[g][u]      (B = softmax tile registers)
    f32x16 pt[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) pt[dt][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int u = kt * 32 + mfma_row(r, h);
        const float* trow = tok + u * GTP + i32;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const float av = (u < GROWS) ? trow[dt * 32] : 0.f;
          pt[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, sacc[kt][r], pt[dt], 0, 0, 0);
        }
        if ((r & 7) == 7) __builtin_amdgcn_sched_barrier(0);
      }
    // (4) ov^T = Wv.PT^T + bv, (5) op^T = Wp.ov^T + bp   (weights pre-transposed [in][out]: lanes = out rows)
    f32x16 ov[2], op[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int r = 0; r < 16; ++r) { ov[jt][r] = cst[128 + jt * 32 + mfma_row(r, h)]; op[jt][r] = cst[192 + jt * 32 + mfma_row(r, h)]; }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* wrow = wv_t + (dt * 32 + mfma_row(r, h)) * GENE_D + i32;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) ov[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wrow[jt * 32], pt[dt][r], ov[jt], 0, 0, 0);
        if ((r & 7) == 7) __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* wrow = wp_t + (dt * 32 + mfma_row(r, h)) * GENE_D + i32;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) op[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wrow[jt * 32], ov[dt][r], op[jt], 0, 0, 0);
        if ((r & 7) == 7) __builtin_amdgcn_sched_barrier(0);
      }
    // (6) norm2 over the 64 features (rows)
    float ss = 0.f;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int r = 0; r < 16; ++r) ss += op[jt][r] * op[jt][r];
    ss += __shfl_xor(ss, 32, 64);
    const float rstd = 1.0f / sqrtf(ss * (1.0f / GENE_D) + TM_EPS);
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int r = 0; r < 16; ++r) op[jt][r] = cst[256 + jt * 32 + mfma_row(r, h)] * (op[jt][r] * rstd);
    // (7) MLP: y1^T = W1.h^T + b1 (256 rows), tanh-GELU, y2^T = W2.y1^T + b2
    f32x16 y2[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int r = 0; r < 16; ++r) y2[jt][r] = cst[320 + jt * 32 + mfma_row(r, h)];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      f32x16 y1;
#pragma unroll
      for (int r = 0; r < 16; ++r) y1[r] = cst[384 + mt * 32 + mfma_row(r, h)];
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          y1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1_t[(jt * 32 + mfma_row(r, h)) * 256 + mt * 32 + i32], op[jt][r], y1, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 16; ++r) y1[r] = gelu_tanh_f(y1[r]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* wrow = w2_t + (mt * 32 + mfma_row(r, h)) * GENE_D + i32;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) y2[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wrow[jt * 32], y1[r], y2[jt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (gok) {
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) a.out_tok[gene_tok_idx(n, g, jt * 32 + mfma_row(r, h), Gb)] = y2[jt][r];
    }
  }
}

hipError_t launch_gene_attn(const float* rna, int B, int gn, int zs, int G, const GeneW& w, float* out_tok,
                            float* attn_map, int zlo, int zhi, hipStream_t s) {
  // fused MFMA form: the checkpoint geometry (D = 64) with G <= GROWS genes; everything else goes through
  // launch_gene_attn_generic
  if (gn * gn * zs != GENE_D || G > GROWS) return hipErrorInvalidValue;
  GeneArgs a;
  a.rna = rna; a.B = B; a.gn = gn; a.zs = zs; a.G = G; a.w = w;
  a.out_tok = out_tok; a.attn_map = attn_map; a.scratch = out_tok;   // norm2 output staged in-place
  a.zlo = zlo; a.zhi = zhi;
  const size_t lds2 = ((size_t)2 * GROWS * GTP + 640) * sizeof(float);
  static DevOnce attr2;
  if (attr2.need()) {
    hipError_t e = hipFuncSetAttribute((const void*)gene_attn_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr2.mark();
  }
  hipLaunchKernelGGL(gene_attn_mfma_kernel, dim3(B, B >= 256 ? 1 : 2), dim3(256), lds2, s, a);
  return hipGetLastError();
}

// ---- generic form of the gene-gene attention block ----------------------------------------------------
// Any gene count G <= 512 and any hidden size D = gn^2 * rna_slc <= 512 (the other patch_size / rna_slc
// configurations, the 500-gene mice, the 81-gene M2H subset).  Correctness-first VALU kernel: one workgroup per
// patch, the per-patch intermediates live in a global scratch slab (L2 resident), every Linear processes RB rows
// per wave so a weight column is fetched once per RB rows.  Same math as the two kernels above:
// q = Linear(tok), qn = RMSNorm(q)*w, P = softmax(qn.qn^T / D), o = norm2(proj(Wv (P.tok) + bv)), MLP.
#define GG_RB 4
struct GeneGenArgs {
  GeneArgs g;
  int D;
  const int* gidx;          // gene g reads slot gidx[g] of the 500 per slice (null: g)
  float* ws; long ws_stride;
};

// out[r][c] = bias[c] + sum_k xs[r*K + k] * W_t[k*Nout + c] for this wave's RB rows, 64 outputs (lane) at a time
template <typename F>
__device__ __forceinline__ void gg_linear(const float* xs, int K, const float* __restrict__ W_t, const float* __restrict__ bias,
                                          int Nout, int lane, F&& emit) {
  for (int c0 = 0; c0 < Nout; c0 += 64) {
    const int c = c0 + lane;
    const bool ok = c < Nout;
    float acc[GG_RB];
    const float b = ok ? bias[c] : 0.f;
#pragma unroll
    for (int r = 0; r < GG_RB; ++r) acc[r] = b;
    // eight weight loads in flight per lane: the loop is latency-bound on the (L2-resident) weight column otherwise
    int k = 0;
    for (; k + 8 <= K; k += 8) {
      float w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) w[u] = ok ? W_t[(long)(k + u) * Nout + c] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int r = 0; r < GG_RB; ++r) acc[r] = fmaf(xs[r * K + k + u], w[u], acc[r]);
    }
    for (; k < K; ++k) {
      const float w = ok ? W_t[(long)k * Nout + c] : 0.f;
#pragma unroll
      for (int r = 0; r < GG_RB; ++r) acc[r] = fmaf(xs[r * K + k], w, acc[r]);
    }
    emit(c, ok, acc);
  }
}

__global__ __launch_bounds__(256) void gene_attn_generic_kernel(GeneGenArgs ga) {
  const GeneArgs& a = ga.g;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int G = a.G, D = ga.D, Gp = (G + 63) / 64 * 64, Gb = (G + 7) / 8;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n = blockIdx.x;
  const int gg = a.gn * a.gn;
  float* tok = ga.ws + ((long)n * gridDim.y + blockIdx.y) * ga.ws_stride;     // [G][D], later the MLP hidden [G][4D]
  float* qn = tok + (long)G * 4 * D;                // [G][D]
  float* qnT = qn + (long)G * D;                    // [D][Gp]
  float* pt = qnT + (long)D * Gp;                   // [G][D]  P.tok, then norm2 output
  float* xs = sm + wv * GG_RB * (4 * D + Gp);       // per wave: RB rows x up to 4D inputs, then RB softmax rows [Gp]
  float* pr = xs + GG_RB * 4 * D;
  const long rbase = (long)n * gg * a.zs * 500;
  for (int i = tid; i < G * D; i += 256) {
    const int g = i / D, d = i - g * D;
    const int z = d / gg, hw = d - z * gg;
    float v = 0.f;
    if (z >= a.zlo && z < a.zhi) v = a.rna[rbase + ((long)hw * a.zs + z) * 500 + (ga.gidx ? ga.gidx[g] : g)];
    tok[i] = v;
  }
  for (int i = tid; i < D * (Gp - G); i += 256) {   // padded key columns of qnT
    const int d = i / (Gp - G), u = G + i - d * (Gp - G);
    qnT[(long)d * Gp + u] = 0.f;
  }
  __syncthreads();
  const float inv_d = 1.0f / (float)D;
  // ---- q = Linear(tok); qn = RMSNorm(q) * w ----
  for (int g0 = wv * GG_RB; g0 < G; g0 += 4 * GG_RB) {
    for (int i = lane; i < GG_RB * D; i += 64) { const int r = i / D; xs[i] = (g0 + r < G) ? tok[(long)(g0 + r) * D + i - r * D] : 0.f; }
    __builtin_amdgcn_wave_barrier();
    float ss[GG_RB] = {0.f, 0.f, 0.f, 0.f};
    gg_linear(xs, D, a.w.wq_t, a.w.bq, D, lane, [&](int c, bool ok, const float (&acc)[GG_RB]) {
#pragma unroll
      for (int r = 0; r < GG_RB; ++r) {
        if (ok && g0 + r < G) qn[(long)(g0 + r) * D + c] = acc[r];
        ss[r] += ok ? acc[r] * acc[r] : 0.f;
      }
    });
#pragma unroll
    for (int r = 0; r < GG_RB; ++r) ss[r] = 1.0f / sqrtf(wave_sum(ss[r]) * inv_d + TM_EPS);
    __builtin_amdgcn_wave_barrier();
    for (int c = lane; c < D; c += 64) {
      const float w = a.w.qnorm[c];
#pragma unroll
      for (int r = 0; r < GG_RB; ++r)
        if (g0 + r < G) {
          const float v = w * (qn[(long)(g0 + r) * D + c] * ss[r]);
          qn[(long)(g0 + r) * D + c] = v;
          qnT[(long)c * Gp + g0 + r] = v;
        }
    }
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  // ---- P = softmax(qn.qn^T / D), RB query rows per wave at a time; pt = P.tok ----
  // (gridDim.y workgroups share a patch's rows of this pass and of the MLP pass; each has its own scratch slab)
  const int NJ = Gp / 64;                             // <= 8 key chunks per lane
  const int row0 = (wv + 4 * blockIdx.y) * GG_RB, rstep = 4 * GG_RB * gridDim.y;
  for (int g0 = row0; g0 < G; g0 += rstep) {
    for (int i = lane; i < GG_RB * D; i += 64) { const int r = i / D; xs[i] = qn[(long)min(g0 + r, G - 1) * D + i - r * D]; }
    __builtin_amdgcn_wave_barrier();
    float lg[8][GG_RB];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < GG_RB; ++r) lg[j][r] = 0.f;
#pragma unroll 2
    for (int d = 0; d < D; ++d) {
      const float* kr = qnT + (long)d * Gp + lane;
      float qv[GG_RB];
#pragma unroll
      for (int r = 0; r < GG_RB; ++r) qv[r] = xs[r * D + d];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < NJ) {
          const float kv = kr[64 * j];
#pragma unroll
          for (int r = 0; r < GG_RB; ++r) lg[j][r] = fmaf(qv[r], kv, lg[j][r]);
        }
    }
#pragma unroll
    for (int r = 0; r < GG_RB; ++r) {
      float m = -INFINITY;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < NJ) { lg[j][r] *= inv_d; if (lane + 64 * j < G) m = fmaxf(m, lg[j][r]); }   // (q*scale).(k)*scale, scale = D^-1/2
      m = wave_max(m);
      float ssum = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < NJ) { lg[j][r] = (lane + 64 * j < G) ? expf(lg[j][r] - m) : 0.f; ssum += lg[j][r]; }
      ssum = wave_sum(ssum);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < NJ) {
          const int u = lane + 64 * j;
          const float p = lg[j][r] / ssum;
          pr[r * Gp + u] = p;
          if (a.attn_map && u < G && g0 + r < G) a.attn_map[((long)n * G + g0 + r) * G + u] = p;
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (a.out_tok) {
      for (int c = lane; c < D; c += 64) {
        float acc[GG_RB];
#pragma unroll
        for (int r = 0; r < GG_RB; ++r) acc[r] = 0.f;
#pragma unroll 4
        for (int u = 0; u < G; ++u) {
          const float tv = tok[(long)u * D + c];
#pragma unroll
          for (int r = 0; r < GG_RB; ++r) acc[r] = fmaf(pr[r * Gp + u], tv, acc[r]);
        }
#pragma unroll
        for (int r = 0; r < GG_RB; ++r) if (g0 + r < G) pt[(long)(g0 + r) * D + c] = acc[r];
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (!a.out_tok) return;
  __syncthreads();
  // ---- ov = Wv.pt + bv; op = Wp.ov + bp; norm2; MLP ----
  float* hrow = tok;                                  // [G][4D] MLP hidden: tok is dead (every wave is past the P.tok pass)
  for (int g0 = row0; g0 < G; g0 += rstep) {
    float* x0 = xs;                                   // [RB][D]
    float* x1 = xs + GG_RB * D;                       // [RB][D]
    for (int i = lane; i < GG_RB * D; i += 64) { const int r = i / D; x0[i] = (g0 + r < G) ? pt[(long)(g0 + r) * D + i - r * D] : 0.f; }
    __builtin_amdgcn_wave_barrier();
    gg_linear(x0, D, a.w.wv_t, a.w.bv, D, lane, [&](int c, bool ok, const float (&acc)[GG_RB]) {
#pragma unroll
      for (int r = 0; r < GG_RB; ++r) if (ok) x1[r * D + c] = acc[r];
    });
    __builtin_amdgcn_wave_barrier();
    float ss[GG_RB] = {0.f, 0.f, 0.f, 0.f};
    gg_linear(x1, D, a.w.wp_t, a.w.bp, D, lane, [&](int c, bool ok, const float (&acc)[GG_RB]) {
#pragma unroll
      for (int r = 0; r < GG_RB; ++r) { if (ok) x0[r * D + c] = acc[r]; ss[r] += ok ? acc[r] * acc[r] : 0.f; }
    });
#pragma unroll
    for (int r = 0; r < GG_RB; ++r) ss[r] = 1.0f / sqrtf(wave_sum(ss[r]) * inv_d + TM_EPS);
    __builtin_amdgcn_wave_barrier();
    for (int c = lane; c < D; c += 64) {              // norm2 (x1 <- normalised rows)
      const float w = a.w.norm2[c];
#pragma unroll
      for (int r = 0; r < GG_RB; ++r) x1[r * D + c] = w * (x0[r * D + c] * ss[r]);
    }
    __builtin_amdgcn_wave_barrier();
    gg_linear(x1, D, a.w.w1_t, a.w.b1, 4 * D, lane, [&](int c, bool ok, const float (&acc)[GG_RB]) {
#pragma unroll
      for (int r = 0; r < GG_RB; ++r) if (ok && g0 + r < G) hrow[(long)(g0 + r) * 4 * D + c] = gelu_tanh_f(acc[r]);
    });
    __threadfence_block();                            // the hidden rows are re-read by other lanes of this wave
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < GG_RB * 4 * D; i += 64) { const int r = i / (4 * D); xs[i] = (g0 + r < G) ? hrow[(long)(g0 + r) * 4 * D + i - r * 4 * D] : 0.f; }
    __builtin_amdgcn_wave_barrier();
    gg_linear(xs, 4 * D, a.w.w2_t, a.w.b2, D, lane, [&](int c, bool ok, const float (&acc)[GG_RB]) {
#pragma unroll
      for (int r = 0; r < GG_RB; ++r)
        if (ok && g0 + r < G) a.out_tok[(((long)n * Gb + ((g0 + r) >> 3)) * D + c) * 8 + ((g0 + r) & 7)] = acc[r];
    });
    __builtin_amdgcn_wave_barrier();
  }
}

int gene_generic_split(int B) { return B >= 512 ? 1 : (B >= 256 ? 2 : 4); }   // workgroups per patch (rows of the heavy passes)

size_t gene_generic_ws_floats(int G, int D) {                                   // per (patch, share) slab
  const size_t Gp = (size_t)(G + 63) / 64 * 64;
  // tok [G][D] (reused as the MLP hidden [G][4D]) + qn [G][D] + qnT [D][Gp] + pt [G][D]
  return (size_t)G * 4 * D + (size_t)G * D + (size_t)D * Gp + (size_t)G * D;
}

hipError_t launch_gene_attn_generic(const float* rna, int B, int gn, int zs, int G, int D, const GeneW& w, const int* gidx,
                                    float* out_tok, float* attn_map, int zlo, int zhi, float* ws, hipStream_t s) {
  if (G < 1 || G > 512 || D < 1 || D > 512 || gn * gn * zs != D) return hipErrorInvalidValue;
  GeneGenArgs ga;
  ga.g.rna = rna; ga.g.B = B; ga.g.gn = gn; ga.g.zs = zs; ga.g.G = G; ga.g.w = w;
  ga.g.out_tok = out_tok; ga.g.attn_map = attn_map; ga.g.scratch = nullptr; ga.g.zlo = zlo; ga.g.zhi = zhi;
  ga.D = D; ga.gidx = gidx; ga.ws = ws; ga.ws_stride = (long)gene_generic_ws_floats(G, D);
  const int Gp = (G + 63) / 64 * 64;
  const size_t lds = (size_t)4 * GG_RB * (4 * D + Gp) * sizeof(float);
  static DevOnce attr_set;
  if (attr_set.need()) {
    hipError_t e = hipFuncSetAttribute((const void*)gene_attn_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set.mark();
  }
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(gene_attn_generic_kernel, dim3(B, gene_generic_split(B)), dim3(256), lds, s, ga);
  return hipGetLastError();
}

// rna_h[:, :, 1:-1] of the attention-map model (model/unet_attn.py:173): [B][G][zs-2][gn][gn]
__global__ void rna_mid_kernel(const float* rna, int B, int gn, int zs, int G, float* out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int gg = gn * gn, zm = zs - 2;
  const long tot = (long)B * G * zm * gg;
  if (i >= tot) return;
  const int hw = (int)(i % gg);
  long r = i / gg;
  const int z = (int)(r % zm); r /= zm;
  const int g = (int)(r % G);
  const int n = (int)(r / G);
  out[i] = rna[(long)n * gg * zs * 500 + ((long)hw * zs + (z + 1)) * 500 + g];
}
hipError_t launch_rna_mid(const float* rna, int B, int gn, int zs, int G, float* out, hipStream_t s) {
  const long tot = (long)B * G * (zs - 2) * gn * gn;
  if (tot <= 0) return hipSuccess;
  hipLaunchKernelGGL(rna_mid_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, rna, B, gn, zs, G, out);
  return hipGetLastError();
}

// ==========================================================================================
// Windowed gene-patch cross attention core (Attention.forward with n_h=2, one head:
// model/MBAblocks.py:555-595): per (patch, window) RMSNorm(q), RMSNorm(k) over C,
// softmax(q.k^T / C) . v.  One workgroup per (patch, window); T = tokens per window.
// ==========================================================================================
struct WinArgs {
  const float *q, *k, *v; long q_ns, k_ns, v_ns;
  const float *qw, *kw;
  float* o; long o_ns;
  int C, Z, S;
  long plane;
  uint16_t* o_h; long o_h_ns;
  int kv_ls;      // != 0: k / v live at half the in-plane resolution (token (z, y, x) reads (z, y >> 1, x >> 1)); = log2(S)
};
// token offset inside a k / v patch, and the k / v channel-block plane, from the q-geometry token offset
#define TM_KVOFF(o) (a.kv_ls ? half_res_off((o), a.kv_ls) : (o))

template <int T>
__global__ __launch_bounds__(256) void window_attn_kernel(WinArgs a) {
  constexpr int TT = T / 16;
  constexpr int KC = 16;
  constexpr int PS = T + 4;          // Pt row stride
  extern __shared__ __attribute__((aligned(16))) float sm[];
  int* tokoff = (int*)sm;            // [T]
  float* rq = sm + T;                // [T]
  float* rk = rq + T;                // [T]
  float* qs = rk + T;                // [KC][T]
  float* ks = qs + KC * T;           // [KC][T]
  float* Pt = ks + KC * T;           // [T][PS]   Pt[u][t]
  float* vs = Pt + T * PS;           // [16][128]
  const int tid = threadIdx.x;
  const int n = blockIdx.x >> 2, win = blockIdx.x & 3;
  const int wy = win >> 1, wx = win & 1;
  const int S = a.S, hs = S / 2, C = a.C;
  for (int t = tid; t < T; t += 256) {
    const int z = t / (hs * hs);
    const int r = t - z * hs * hs;
    const int yl = r / hs, xl = r - yl * hs;
    tokoff[t] = ((z * S + wy * hs + yl) * S + wx * hs + xl) * 8;
  }
  __syncthreads();
  const float* qb = a.q + (long)n * a.q_ns;
  const float* kb = a.k + (long)n * a.k_ns;
  const float* vb = a.v + (long)n * a.v_ns;
  const long kvp = a.kv_ls ? a.plane >> 2 : a.plane;             // channel-block plane of k / v
  if (tid < 2 * T) {
    const bool isq = tid < T;
    const int t = isq ? tid : tid - T;
    const float* p = isq ? qb + tokoff[t] : kb + TM_KVOFF(tokoff[t]);
    const long pp = isq ? a.plane : kvp;
    float ss = 0.f;
    for (int cb = 0; cb < C / 8; ++cb) {
      const f32x4 a0 = *(const f32x4*)(p + (long)cb * pp), a1 = *(const f32x4*)(p + (long)cb * pp + 4);
      ss += a0[0] * a0[0] + a0[1] * a0[1] + a0[2] * a0[2] + a0[3] * a0[3] + a1[0] * a1[0] + a1[1] * a1[1] +
            a1[2] * a1[2] + a1[3] * a1[3];
    }
    const float r = 1.0f / sqrtf(ss / (float)C + TM_EPS);
    if (isq) rq[t] = r; else rk[t] = r;
  }
  __syncthreads();
  const int ty = tid >> 4, tx = tid & 15;
  float acc[TT][TT];
#pragma unroll
  for (int i = 0; i < TT; ++i)
#pragma unroll
    for (int j = 0; j < TT; ++j) acc[i][j] = 0.f;
  for (int c0 = 0; c0 < C; c0 += KC) {
    // stage: items = (q|k, token, cblk of 2)
    for (int it = tid; it < 2 * T * 2; it += 256) {
      const int cbi = it & 1;
      const int t = (it >> 1) % T;
      const bool isq = (it >> 1) < T;
      const int cb = c0 / 8 + cbi;
      const float* p = isq ? qb + tokoff[t] + (long)cb * a.plane : kb + TM_KVOFF(tokoff[t]) + (long)cb * kvp;
      const f32x4 a0 = *(const f32x4*)p, a1 = *(const f32x4*)(p + 4);
      const float r = isq ? rq[t] : rk[t];
      const float* nw = (isq ? a.qw : a.kw) + cb * 8;
      float* dst = (isq ? qs : ks) + (cbi * 8) * T + t;
      dst[0 * T] = nw[0] * (a0[0] * r); dst[1 * T] = nw[1] * (a0[1] * r);
      dst[2 * T] = nw[2] * (a0[2] * r); dst[3 * T] = nw[3] * (a0[3] * r);
      dst[4 * T] = nw[4] * (a1[0] * r); dst[5 * T] = nw[5] * (a1[1] * r);
      dst[6 * T] = nw[6] * (a1[2] * r); dst[7 * T] = nw[7] * (a1[3] * r);
    }
    __syncthreads();
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      float av[TT], bv[TT];
#pragma unroll
      for (int i = 0; i < TT; ++i) av[i] = qs[kc * T + ty * TT + i];
#pragma unroll
      for (int j = 0; j < TT; ++j) bv[j] = ks[kc * T + tx * TT + j];
#pragma unroll
      for (int i = 0; i < TT; ++i)
#pragma unroll
        for (int j = 0; j < TT; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
    __syncthreads();
  }
  const float inv_c = 1.0f / (float)C;        // (q*scale) . k * scale, scale = C^-1/2
#pragma unroll
  for (int i = 0; i < TT; ++i)
#pragma unroll
    for (int j = 0; j < TT; ++j) Pt[(tx * TT + j) * PS + ty * TT + i] = acc[i][j] * inv_c;
  __syncthreads();
  if (tid < T) {
    float m = -INFINITY;
    for (int u = 0; u < T; ++u) m = fmaxf(m, Pt[u * PS + tid]);
    float ssum = 0.f;
    for (int u = 0; u < T; ++u) { const float e = expf(Pt[u * PS + tid] - m); Pt[u * PS + tid] = e; ssum += e; }
    for (int u = 0; u < T; ++u) Pt[u * PS + tid] = Pt[u * PS + tid] / ssum;
  }
  __syncthreads();
  float* ob = a.o + (long)n * a.o_ns;
  for (int c0 = 0; c0 < C; c0 += 128) {
    float oa[TT][8];
#pragma unroll
    for (int i = 0; i < TT; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) oa[i][j] = 0.f;
    for (int u0 = 0; u0 < T; u0 += 16) {
      {
        const int uu = tid >> 4, cbi = tid & 15;
        const float* p = vb + TM_KVOFF(tokoff[u0 + uu]) + (long)(c0 / 8 + cbi) * kvp;
        *(f32x4*)(vs + uu * 128 + cbi * 8) = *(const f32x4*)p;
        *(f32x4*)(vs + uu * 128 + cbi * 8 + 4) = *(const f32x4*)(p + 4);
      }
      __syncthreads();
#pragma unroll
      for (int uu = 0; uu < 16; ++uu) {
        float pv[TT];
#pragma unroll
        for (int i = 0; i < TT; ++i) pv[i] = Pt[(u0 + uu) * PS + ty * TT + i];
        const f32x4 v0 = *(const f32x4*)(vs + uu * 128 + tx * 8), v1 = *(const f32x4*)(vs + uu * 128 + tx * 8 + 4);
#pragma unroll
        for (int i = 0; i < TT; ++i) {
          oa[i][0] = fmaf(pv[i], v0[0], oa[i][0]); oa[i][1] = fmaf(pv[i], v0[1], oa[i][1]);
          oa[i][2] = fmaf(pv[i], v0[2], oa[i][2]); oa[i][3] = fmaf(pv[i], v0[3], oa[i][3]);
          oa[i][4] = fmaf(pv[i], v1[0], oa[i][4]); oa[i][5] = fmaf(pv[i], v1[1], oa[i][5]);
          oa[i][6] = fmaf(pv[i], v1[2], oa[i][6]); oa[i][7] = fmaf(pv[i], v1[3], oa[i][7]);
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < TT; ++i) {
      const long eo = tokoff[ty * TT + i] + (long)(c0 / 8 + tx) * a.plane;
      if (a.o_h) {
        typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
        bf16x8 v8;
#pragma unroll
        for (int j = 0; j < 8; ++j) v8[j] = (__bf16)oa[i][j];
        *(bf16x8*)(a.o_h + (long)n * a.o_h_ns + eo) = v8;
      } else {
        float* p = ob + eo;
        *(f32x4*)p = f32x4{oa[i][0], oa[i][1], oa[i][2], oa[i][3]};
        *(f32x4*)(p + 4) = f32x4{oa[i][4], oa[i][5], oa[i][6], oa[i][7]};
      }
    }
  }
}

// ---- MFMA form for T = 128 tokens per window (the resolution-16 AttnBlocks) ---------------------------
// S = Qn.Kn^T and O = P.V both run on v_mfma_f32_32x32x2_f32 (exact fp32); the softmax row reduction is a
// wave-shuffle butterfly over the 32 lanes that hold a row's 32 columns.  Wave w owns query rows [32w, 32w+32).
//   QK^T : A = Q (rows = queries), B = K (cols = keys); both fragments are float4 global loads of 4 channels of
//          one token (lanes 0-31: channels 0-3, lanes 32-63: channels 4-7 of the 8-channel block), q/k RMSNorm
//          weights folded into the K fragment, the two rstd factors applied to the 32x32 result.
//   P.V  : computed as O^T = V^T.P^T so that a lane owns ONE token and 4 consecutive channels per accumulator
//          quad (same coalesced CB8 / bf16 store as the conv epilogue); P goes through LDS row-major, V is
//          staged transposed ([channel][token]) 32 channels at a time.
struct WinLds {
  static constexpr int PS = 132;                 // row pitch (floats): b128 fragment reads conflict free
  static constexpr int FLOATS = 128 * PS + 32 * PS + 3 * 128 + 512;
};

__global__ __launch_bounds__(256) void window_attn_mfma_kernel(WinArgs a) {
  constexpr int T = 128, PS = WinLds::PS;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* P = sm;                       // [T][PS]
  float* Vt = P + T * PS;              // [32][PS]
  float* rq = Vt + 32 * PS;            // [T]
  float* rk = rq + T;                  // [T]
  int* tokoff = (int*)(rk + T);        // [T]
  float* w2 = (float*)(tokoff + T);    // [C] q_norm.weight * k_norm.weight (C <= 512)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int i32 = lane & 31, h = lane >> 5;
  const int n = blockIdx.x >> 2, win = blockIdx.x & 3;
  const int wy = win >> 1, wx = win & 1;
  const int S = a.S, hs = S / 2, C = a.C;
  if (tid < T) {
    const int z = tid / (hs * hs);
    const int r = tid - z * hs * hs;
    const int yl = r / hs, xl = r - yl * hs;
    tokoff[tid] = ((z * S + wy * hs + yl) * S + wx * hs + xl) * 8;
  }
  for (int c = tid; c < C; c += 256) w2[c] = a.qw[c] * a.kw[c];
  __syncthreads();
  const float* qb = a.q + (long)n * a.q_ns;
  const float* kb = a.k + (long)n * a.k_ns;
  const float* vb = a.v + (long)n * a.v_ns;
  const long kvp = a.kv_ls ? a.plane >> 2 : a.plane;             // channel-block plane of k / v
  {
    const bool isq = tid < T;
    const int t = isq ? tid : tid - T;
    const float* p = isq ? qb + tokoff[t] : kb + TM_KVOFF(tokoff[t]);
    const long pp = isq ? a.plane : kvp;
    float ss = 0.f;
    for (int cb = 0; cb < C / 8; ++cb) {
      const f32x4 a0 = *(const f32x4*)(p + (long)cb * pp), a1 = *(const f32x4*)(p + (long)cb * pp + 4);
      ss += a0[0] * a0[0] + a0[1] * a0[1] + a0[2] * a0[2] + a0[3] * a0[3] + a1[0] * a1[0] + a1[1] * a1[1] +
            a1[2] * a1[2] + a1[3] * a1[3];
    }
    const float r = 1.0f / sqrtf(ss / (float)C + TM_EPS);
    if (isq) rq[t] = r; else rk[t] = r;
  }
  __syncthreads();

  // ---- S = Q.K^T ----
  f32x16 acc[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
  const float* qp = qb + tokoff[wv * 32 + i32] + 4 * h;
  const float* kp[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) kp[ct] = kb + TM_KVOFF(tokoff[ct * 32 + i32]) + 4 * h;
  // fragments of channel block cb+1 are in flight while block cb's 16 MFMAs issue (one workgroup per CU: nothing
  // else would hide the L2 round trip)
  f32x4 qn = *(const f32x4*)qp, kn[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) kn[ct] = *(const f32x4*)kp[ct];
  for (int cb = 0; cb < C / 8; ++cb) {
    const f32x4 qf = qn;
    const f32x4 wf = *(const f32x4*)(w2 + cb * 8 + 4 * h);
    f32x4 kf[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) kf[ct] = kn[ct] * wf;
    if (cb + 1 < C / 8) {
      const long po = (long)(cb + 1) * a.plane, pk = (long)(cb + 1) * kvp;
      qn = *(const f32x4*)(qp + po);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) kn[ct] = *(const f32x4*)(kp[ct] + pk);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[kk], kf[ct][kk], acc[ct], 0, 0, 0);
  }
  // ---- scale, softmax over keys (columns = lanes of this 32-lane half x 4 column tiles) ----
  const float inv_c = 1.0f / (float)C;                   // (q*scale).k*scale, scale = C^-1/2 (MBAblocks.py:571-577)
  float rkc[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) rkc[ct] = rk[ct * 32 + i32] * inv_c;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;      // C/D layout of the 32x32 MFMA
    const float rqr = rq[wv * 32 + row];
    float m = -INFINITY;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) { acc[ct][r] *= rqr * rkc[ct]; m = fmaxf(m, acc[ct][r]); }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float ssum = 0.f;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) { acc[ct][r] = expf(acc[ct][r] - m); ssum += acc[ct][r]; }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) ssum += __shfl_xor(ssum, o, 64);
    const float inv = 1.0f / ssum;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) P[(wv * 32 + row) * PS + ct * 32 + i32] = acc[ct][r] * inv;
  }
  // ---- O^T = V^T.P^T, 32 channels at a time ----
  const float* pfrag = P + (wv * 32 + i32) * PS + 4 * h;          // B operand: P[t_j][u0 + 4h ..]
  const float* vfrag = Vt + i32 * PS + 4 * h;                      // A operand: Vt[c_i][u0 + 4h ..]
  const int myoff = tokoff[wv * 32 + i32];
  // V chunk staging: thread owns items (u, cbi) = (tid & 127, tid >> 7) and (tid & 127, 2 + (tid >> 7)); the next
  // chunk's two 32-byte pieces are loaded into registers before the current chunk's MFMAs
  const int su = tid & (T - 1), scb = tid >> 7;
  const float* vsrc = vb + TM_KVOFF(tokoff[su]) + (long)scb * kvp;
  f32x4 vr[4];
  vr[0] = *(const f32x4*)vsrc; vr[1] = *(const f32x4*)(vsrc + 4);
  vr[2] = *(const f32x4*)(vsrc + 2 * kvp); vr[3] = *(const f32x4*)(vsrc + 2 * kvp + 4);
  for (int c0 = 0; c0 < C; c0 += 32) {
    __syncthreads();                                               // Vt free (and, first time, P complete)
#pragma unroll
    for (int half = 0; half < 2; ++half) {                         // transposed store [channel][token]
      float* d = Vt + ((scb + 2 * half) * 8) * PS + su;
      const f32x4 v0 = vr[2 * half], v1 = vr[2 * half + 1];
      d[0 * PS] = v0[0]; d[1 * PS] = v0[1]; d[2 * PS] = v0[2]; d[3 * PS] = v0[3];
      d[4 * PS] = v1[0]; d[5 * PS] = v1[1]; d[6 * PS] = v1[2]; d[7 * PS] = v1[3];
    }
    if (c0 + 32 < C) {
      const float* p = vsrc + (long)((c0 + 32) / 8) * kvp;
      vr[0] = *(const f32x4*)p; vr[1] = *(const f32x4*)(p + 4);
      vr[2] = *(const f32x4*)(p + 2 * kvp); vr[3] = *(const f32x4*)(p + 2 * kvp + 4);
    }
    __syncthreads();
    f32x16 oc;
#pragma unroll
    for (int r = 0; r < 16; ++r) oc[r] = 0.f;
#pragma unroll 4
    for (int u0 = 0; u0 < T; u0 += 8) {
      const f32x4 af = *(const f32x4*)(vfrag + u0), bf = *(const f32x4*)(pfrag + u0);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) oc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk], bf[kk], oc, 0, 0, 0);
    }
    // lane = token (wv*32 + i32); accumulator quad g = channels c0 + 8g + 4h .. +3
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const long eo = myoff + (long)(c0 / 8 + g) * a.plane + 4 * h;
      if (a.o_h) {
        typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
        bf16x4_t ob;
#pragma unroll
        for (int j = 0; j < 4; ++j) ob[j] = (__bf16)oc[4 * g + j];
        *(bf16x4_t*)(a.o_h + (long)n * a.o_h_ns + eo) = ob;
      } else {
        *(f32x4*)(a.o + (long)n * a.o_ns + eo) = f32x4{oc[4 * g + 0], oc[4 * g + 1], oc[4 * g + 2], oc[4 * g + 3]};
      }
    }
  }
}



// ---- fp32 MFMA form for long windows (T = 256 / 512: z_size 4 / 8 at the resolution-16 attention blocks) ----------------
// One workgroup = 128 queries of one window; the keys are walked in blocks of 128 with an exact two-pass softmax
// (pass 1: every query's max / sum over all keys; pass 2: per block P = exp(s - m) / l into LDS, O^T += V_blk^T.P_blk^T).
// S^T = K.Q^T on v_mfma_f32_32x32x2_f32 (A = K: rows = keys, B = Q with q_norm.w * k_norm.w folded in), so a lane owns
// one query: the softmax reductions are in-lane plus one exchange with lane ^ 32.  C <= 256.
struct WinLongLds {
  static constexpr int PS = 132;                 // row pitch (floats) of the P block and of a V^T row
};

template <int T>
__global__ __launch_bounds__(256) void window_attn_mfma_long_kernel(WinArgs a) {
  constexpr int PS = WinLongLds::PS, KBN = T / 128, QBN = T / 128;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* P = sm;                       // [128 queries][PS]
  float* Vt = P + 128 * PS;            // [32 channels][PS]
  float* rq = Vt + 32 * PS;            // [128]
  float* rk = rq + 128;                // [T]
  int* tokoff = (int*)(rk + T);        // [T]
  float* w2 = (float*)(tokoff + T);    // [C]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int i32 = lane & 31, h = lane >> 5;
  const int qblk = blockIdx.x % QBN, win = (blockIdx.x / QBN) & 3, n = blockIdx.x / (4 * QBN);
  const int wy = win >> 1, wx = win & 1;
  const int S = a.S, hs = S / 2, C = a.C, Cb = C / 8;
  for (int t = tid; t < T; t += 256) {
    const int z = t / (hs * hs);
    const int r = t - z * hs * hs;
    const int yl = r / hs, xl = r - yl * hs;
    tokoff[t] = ((z * S + wy * hs + yl) * S + wx * hs + xl) * 8;
  }
  for (int c = tid; c < C; c += 256) w2[c] = a.qw[c] * a.kw[c];
  __syncthreads();
  const float* qb = a.q + (long)n * a.q_ns;
  const float* kb = a.k + (long)n * a.k_ns;
  const float* vb = a.v + (long)n * a.v_ns;
  for (int i = tid; i < 128 + T; i += 256) {
    const bool isq = i < 128;
    const int t = isq ? qblk * 128 + i : i - 128;
    const float* p = (isq ? qb : kb) + tokoff[t];
    float ss = 0.f;
    for (int cb = 0; cb < Cb; ++cb) {
      const f32x4 a0 = *(const f32x4*)(p + (long)cb * a.plane), a1 = *(const f32x4*)(p + (long)cb * a.plane + 4);
      ss += a0[0] * a0[0] + a0[1] * a0[1] + a0[2] * a0[2] + a0[3] * a0[3] + a1[0] * a1[0] + a1[1] * a1[1] +
            a1[2] * a1[2] + a1[3] * a1[3];
    }
    const float r = 1.0f / sqrtf(ss / (float)C + TM_EPS);
    if (isq) rq[i] = r; else rk[t] = r;
  }
  __syncthreads();

  const int qloc = wv * 32 + i32;
  const int qoff = tokoff[qblk * 128 + qloc];
  const float sq = rq[qloc] / (float)C;                  // (q*scale).(k*scale), scale = C^-1/2 (MBAblocks.py:571-577)
  const float* qp = qb + qoff + 4 * h;
  f32x16 acc[4];
  auto s_block = [&](int kblk) {
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
    const float* kp[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) kp[ct] = kb + tokoff[kblk * 128 + ct * 32 + i32] + 4 * h;
    f32x4 qn = *(const f32x4*)qp, kn[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) kn[ct] = *(const f32x4*)kp[ct];
    for (int cb = 0; cb < Cb; ++cb) {
      const f32x4 qf = qn * *(const f32x4*)(w2 + cb * 8 + 4 * h);
      f32x4 kf[4];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) kf[ct] = kn[ct];
      if (cb + 1 < Cb) {
        const long po = (long)(cb + 1) * a.plane;
        qn = *(const f32x4*)(qp + po);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) kn[ct] = *(const f32x4*)(kp[ct] + po);
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[ct][kk], qf[kk], acc[ct], 0, 0, 0);
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ct][r] *= sq * rk[kblk * 128 + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
  };
  float m = -INFINITY, l = 0.f;
  for (int kblk = 0; kblk < KBN; ++kblk) {
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
  f32x16 oc[8];
#pragma unroll
  for (int tI = 0; tI < 8; ++tI)
#pragma unroll
    for (int r = 0; r < 16; ++r) oc[tI][r] = 0.f;
  const int su = tid & 127, scb = tid >> 7;              // V staging: key su of the block, channel blocks scb and scb + 2
  for (int kblk = 0; kblk < KBN; ++kblk) {
    s_block(kblk);
    float* prow = P + qloc * PS;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 pk;
#pragma unroll
        for (int j = 0; j < 4; ++j) pk[j] = expf(acc[ct][4 * g + j] - m) * inv;
        *(f32x4*)(prow + ct * 32 + 8 * g + 4 * h) = pk;
      }
    const float* vsrc = vb + tokoff[kblk * 128 + su] + (long)scb * a.plane;
    const float* pfrag = P + qloc * PS + 4 * h;          // B operand: P[query][key0 + 4h ..]
    const float* vfrag = Vt + i32 * PS + 4 * h;          // A operand: Vt[channel][key0 + 4h ..]
    for (int c0 = 0; c0 < C; c0 += 32) {
      __syncthreads();                                   // V^T free; (first time) the P rows are complete
      {
        const float* p0 = vsrc + (long)(c0 / 8) * a.plane;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const f32x4 v0 = *(const f32x4*)(p0 + (long)(2 * half) * a.plane), v1 = *(const f32x4*)(p0 + (long)(2 * half) * a.plane + 4);
          float* d = Vt + ((scb + 2 * half) * 8) * PS + su;
          d[0 * PS] = v0[0]; d[1 * PS] = v0[1]; d[2 * PS] = v0[2]; d[3 * PS] = v0[3];
          d[4 * PS] = v1[0]; d[5 * PS] = v1[1]; d[6 * PS] = v1[2]; d[7 * PS] = v1[3];
        }
      }
      __syncthreads();
      const int tI = c0 / 32;
#pragma unroll
      for (int cc = 0; cc < 8; ++cc)
        if (cc == tI) {
#pragma unroll 4
          for (int u0 = 0; u0 < 128; u0 += 8) {
            const f32x4 af = *(const f32x4*)(vfrag + u0), bf = *(const f32x4*)(pfrag + u0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) oc[cc] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk], bf[kk], oc[cc], 0, 0, 0);
          }
        }
    }
    __syncthreads();                                     // P block consumed before the next block overwrites it
  }
#pragma unroll
  for (int tI = 0; tI < 8; ++tI) {
    if (tI * 32 >= C) break;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *(f32x4*)(a.o + (long)n * a.o_ns + qoff + (long)(tI * 4 + g) * a.plane + 4 * h) =
          f32x4{oc[tI][4 * g + 0], oc[tI][4 * g + 1], oc[tI][4 * g + 2], oc[tI][4 * g + 3]};
  }
}

// ---- generic windowed attention core: any window size T = Z*(S/2)^2 <= 512, C <= 512 (fp32, VALU) ------------------
// The other patch_size / rna_slc configurations (T = 8 ... 512).  One workgroup per (patch, window); a wave works on
// WQ queries at a time so that every K / V fragment it loads feeds WQ dot products / accumulations: lanes = keys for the
// logits, lanes = channel blocks for P.V.  < 0.5 % of the FLOPs; correctness first.
#define WQ 4
__global__ __launch_bounds__(256) void window_attn_generic_kernel(WinArgs a, int T) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int C = a.C, Cb = C / 8;
  int* tokoff = (int*)sm;                  // [T]
  float* rq = sm + T;                      // [T]
  float* rk = rq + T;                      // [T]
  float* w2 = rk + T;                      // [C]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float* qrow = w2 + C + wv * WQ * (C + T);   // [WQ][C]
  float* prow = qrow + WQ * C;                // [WQ][T]
  const int n = blockIdx.x >> 2, win = blockIdx.x & 3;
  const int wy = win >> 1, wx = win & 1;
  const int S = a.S, hs = S / 2;
  for (int t = tid; t < T; t += 256) {
    const int z = t / (hs * hs);
    const int r = t - z * hs * hs;
    const int yl = r / hs, xl = r - yl * hs;
    tokoff[t] = ((z * S + wy * hs + yl) * S + wx * hs + xl) * 8;
  }
  for (int c = tid; c < C; c += 256) w2[c] = a.qw[c] * a.kw[c];
  __syncthreads();
  const float* qb = a.q + (long)n * a.q_ns;
  const float* kb = a.k + (long)n * a.k_ns;
  const float* vb = a.v + (long)n * a.v_ns;
  for (int i = tid; i < 2 * T; i += 256) {
    const bool isq = i < T;
    const int t = isq ? i : i - T;
    const float* p = (isq ? qb : kb) + tokoff[t];
    float ss = 0.f;
    for (int cb = 0; cb < Cb; ++cb) {
      const f32x4 a0 = *(const f32x4*)(p + (long)cb * a.plane), a1 = *(const f32x4*)(p + (long)cb * a.plane + 4);
      ss += a0[0] * a0[0] + a0[1] * a0[1] + a0[2] * a0[2] + a0[3] * a0[3] + a1[0] * a1[0] + a1[1] * a1[1] +
            a1[2] * a1[2] + a1[3] * a1[3];
    }
    const float r = 1.0f / sqrtf(ss / (float)C + TM_EPS);
    if (isq) rq[t] = r; else rk[t] = r;
  }
  __syncthreads();
  const int NJ = (T + 63) / 64;            // <= 8 key chunks per lane
  for (int t0 = wv * WQ; t0 < T; t0 += 4 * WQ) {
    // (q*scale).(k*scale), scale = C^-1/2 (MBAblocks.py:571-577); queries past T replicate the last one (never stored)
    for (int i = lane; i < WQ * C; i += 64) {
      const int r = i / C, c = i - r * C;
      const int t = min(t0 + r, T - 1);
      qrow[i] = qb[tokoff[t] + (long)(c >> 3) * a.plane + (c & 7)] * w2[c] * (rq[t] / (float)C);
    }
    __builtin_amdgcn_wave_barrier();
    float lg[8][WQ];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int u = lane + 64 * j;
      const bool ok = j < NJ && u < T;
      float acc[WQ];
#pragma unroll
      for (int r = 0; r < WQ; ++r) acc[r] = 0.f;
      if (ok) {
        const float* kp = kb + tokoff[u];
#pragma unroll 4
        for (int cb = 0; cb < Cb; ++cb) {
          const f32x4 k0 = *(const f32x4*)(kp + (long)cb * a.plane), k1 = *(const f32x4*)(kp + (long)cb * a.plane + 4);
#pragma unroll
          for (int r = 0; r < WQ; ++r) {
            const float* qc = qrow + r * C + cb * 8;
            acc[r] += qc[0] * k0[0] + qc[1] * k0[1] + qc[2] * k0[2] + qc[3] * k0[3] + qc[4] * k1[0] + qc[5] * k1[1] +
                      qc[6] * k1[2] + qc[7] * k1[3];
          }
        }
      }
      const float rku = ok ? rk[u] : 0.f;
#pragma unroll
      for (int r = 0; r < WQ; ++r) lg[j][r] = ok ? acc[r] * rku : -INFINITY;
    }
#pragma unroll
    for (int r = 0; r < WQ; ++r) {
      float m = -INFINITY;
#pragma unroll
      for (int j = 0; j < 8; ++j) m = fmaxf(m, lg[j][r]);
      m = wave_max(m);
      float ssum = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) { lg[j][r] = (lg[j][r] == -INFINITY) ? 0.f : expf(lg[j][r] - m); ssum += lg[j][r]; }
      ssum = wave_sum(ssum);
      const float inv = 1.0f / ssum;
#pragma unroll
      for (int j = 0; j < 8; ++j) { const int u = lane + 64 * j; if (j < NJ && u < T) prow[r * T + u] = lg[j][r] * inv; }
    }
    __builtin_amdgcn_wave_barrier();
    for (int cb = lane; cb < Cb; cb += 64) {
      float o[WQ][8];
#pragma unroll
      for (int r = 0; r < WQ; ++r)
#pragma unroll
        for (int j = 0; j < 8; ++j) o[r][j] = 0.f;
      const float* vp = vb + (long)cb * a.plane;
#pragma unroll 4
      for (int u = 0; u < T; ++u) {
        const f32x4 v0 = *(const f32x4*)(vp + tokoff[u]), v1 = *(const f32x4*)(vp + tokoff[u] + 4);
#pragma unroll
        for (int r = 0; r < WQ; ++r) {
          const float p = prow[r * T + u];
          o[r][0] = fmaf(p, v0[0], o[r][0]); o[r][1] = fmaf(p, v0[1], o[r][1]); o[r][2] = fmaf(p, v0[2], o[r][2]);
          o[r][3] = fmaf(p, v0[3], o[r][3]); o[r][4] = fmaf(p, v1[0], o[r][4]); o[r][5] = fmaf(p, v1[1], o[r][5]);
          o[r][6] = fmaf(p, v1[2], o[r][6]); o[r][7] = fmaf(p, v1[3], o[r][7]);
        }
      }
#pragma unroll
      for (int r = 0; r < WQ; ++r) {
        if (t0 + r >= T) break;
        float* op = a.o + (long)n * a.o_ns + tokoff[t0 + r] + (long)cb * a.plane;
        *(f32x4*)op = f32x4{o[r][0], o[r][1], o[r][2], o[r][3]};
        *(f32x4*)(op + 4) = f32x4{o[r][4], o[r][5], o[r][6], o[r][7]};
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

template <int T>
static hipError_t launch_win(const WinArgs& a, int N, hipStream_t s) {
  const size_t lds = ((size_t)3 * T + 2 * 16 * T + (size_t)T * (T + 4) + 16 * 128) * sizeof(float);
  static DevOnce attr_set;
  if (attr_set.need()) {
    hipError_t e = hipFuncSetAttribute((const void*)window_attn_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set.mark();
  }
  hipLaunchKernelGGL(window_attn_kernel<T>, dim3(N * 4), dim3(256), lds, s, a);
  return hipGetLastError();
}

hipError_t launch_window_attn(const TV& q, const TV& k, const TV& v, const float* qnorm_w, const float* knorm_w,
                              TV o, hipStream_t s, uint16_t* o_h, long o_h_nstride) {
  WinArgs a;
  a.o_h = o_h; a.o_h_ns = o_h_nstride;
  a.q = q.p; a.k = k.p; a.v = v.p; a.q_ns = q.nstride; a.k_ns = k.nstride; a.v_ns = v.nstride;
  a.qw = qnorm_w; a.kw = knorm_w; a.o = o.p; a.o_ns = o.nstride;
  a.C = q.Cb * 8; a.Z = q.Z; a.S = q.H; a.plane = q.plane();
  if (a.C % 8 || q.H != q.W || (q.H & 1)) return hipErrorInvalidValue;
  const int T = q.Z * (q.H / 2) * (q.H / 2);
  a.kv_ls = 0;
  if (k.H != q.H) {     // k / v at half the in-plane resolution: the T = 128 MFMA and the T = 32 kernels only, S a power of two
    int ls = 0;
    while ((1 << ls) < q.H) ++ls;
    if (k.H * 2 != q.H || v.H != k.H || (1 << ls) != q.H || ls < 2 || a.C % 128 || !((T == 128 && a.C <= 512) || T == 32))
      return hipErrorInvalidValue;
    a.kv_ls = ls;
  }
  if ((T == 256 || T == 512) && a.C <= 256 && !o_h) {   // long windows: key-blocked fp32 MFMA form
    const size_t lds = ((size_t)(128 + 32) * WinLongLds::PS + 128 + 2 * T + a.C) * sizeof(float);
#define TM_LAUNCHWL(T_)                                                                                              \
  do {                                                                                                               \
    static DevOnce lattr;                                                                                       \
    if (lattr.need()) {                                                                                                    \
      hipError_t e = hipFuncSetAttribute((const void*)window_attn_mfma_long_kernel<T_>,                              \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                    \
      if (e != hipSuccess) return e;                                                                                 \
      lattr.mark();                                                                                                  \
    }                                                                                                                \
    hipLaunchKernelGGL(window_attn_mfma_long_kernel<T_>, dim3(q.N * 4 * (T_ / 128)), dim3(256), lds, s, a);          \
  } while (0)
    if (T == 256) TM_LAUNCHWL(256); else TM_LAUNCHWL(512);
#undef TM_LAUNCHWL
    return hipGetLastError();
  }
  if ((T != 128 && T != 32) || a.C % 128) {            // the other configurations: generic kernel (fp32 output only)
    if (T > 512 || a.C > 512 || o_h) return hipErrorInvalidValue;
    const size_t lds = ((size_t)3 * T + a.C + 4 * WQ * (a.C + T)) * sizeof(float);
    static DevOnce gattr;
    if (gattr.need()) {
      hipError_t e = hipFuncSetAttribute((const void*)window_attn_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
      gattr.mark();
    }
    hipLaunchKernelGGL(window_attn_generic_kernel, dim3(q.N * 4), dim3(256), lds, s, a, T);
    return hipGetLastError();
  }
  if (T == 128 && a.C <= 512) {
    static DevOnce attr_set;
    const size_t lds = (size_t)WinLds::FLOATS * sizeof(float);
    if (attr_set.need()) {
      hipError_t e = hipFuncSetAttribute((const void*)window_attn_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
      attr_set.mark();
    }
    hipLaunchKernelGGL(window_attn_mfma_kernel, dim3(q.N * 4), dim3(256), lds, s, a);
    return hipGetLastError();
  }
  if (T == 32) return launch_win<32>(a, q.N, s);
  return hipErrorInvalidValue;
}

}  // namespace tmk
