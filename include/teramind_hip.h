/* teramind_hip.h -- C ABI of libteramind_hip.so (MI355X / gfx950 only).
 *
 * The reference (CTPLab/Tera-MIND) has no FFI / plugin interface for this path: its hot
 * path is stock PyTorch modules.  This ABI is what a binding for the path would attach to;
 * each entry point cites the reference interface it replaces (paths relative to the
 * reference repository root).  INTEGRATION.md shows the ctypes stub a maintainer would add.
 *
 * Conventions
 *   - all tensor pointers are DEVICE pointers owned by the caller (PyTorch-ROCm
 *     allocations); the library borrows them for the duration of the enqueue and never
 *     frees or allocates caller-visible memory.  The only library-owned device memory is
 *     the packed weight arena (freed by tm_model_destroy).
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*); no hidden
 *     device synchronisation, no internal threads.  One host thread per model.
 *   - return value: 0 = OK, negative = TM_ERR_*; text via tm_last_error() (thread local).
 *     No C++ exception crosses the ABI.
 *   - image-like tensors are fp32, contiguous, NCHW as in the reference ("b (s z) h w").
 */
#ifndef TERAMIND_HIP_H
#define TERAMIND_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TM_ABI_VERSION 1

enum {
  TM_OK = 0,
  TM_ERR_ARG = -1,        /* bad argument / unsupported configuration */
  TM_ERR_STATE = -2,      /* call order violated (e.g. forward before finalize) */
  TM_ERR_KEY = -3,        /* unknown / duplicate / missing state_dict key, or shape mismatch */
  TM_ERR_HIP = -4,        /* a HIP runtime call failed */
  TM_ERR_WORKSPACE = -5   /* workspace too small */
};

/* TM_DTYPE_F32: every kernel computes in fp32 (exact-fp32 MFMA).  TM_DTYPE_BF16: the 3x3x3 convs
 * (95 % of the FLOPs) take bf16 weights and bf16 normalised/activated inputs on the bf16 MFMA with
 * fp32 accumulation; the residual stream, norms, attention and everything else stay fp32.
 * TM_DTYPE_F16: the same kernels with IEEE-half operands -- the arithmetic the reference runs on GPUs
 * (autocast('cuda', fp16), diffusion/base.py:377): same speed as bf16, 3 more mantissa bits, fp16 range.
 * Parameters are always loaded as fp32 and rounded (RNE) when packed. */
enum { TM_DTYPE_F32 = 0, TM_DTYPE_BF16 = 1, TM_DTYPE_F16 = 2 };
enum { TM_SAMPLE_DDPM = 0, TM_SAMPLE_DDIM = 1 };

/* Model configuration.  Replaces BeatGANsUNetConfig as filled by
 * TrainConfig.make_model_conf (config.py:280-326) from prep_config_parm (config_parm.py:5-59). */
typedef struct tm_config {
  int32_t patch_size;     /* image_size / patch_size: 32 | 64 | 128 (config_parm.py:47-55; 64 = the published checkpoint) */
  int32_t rna_slc;        /* len(rna_tpl): 1 | 4 | 8 | 16 (train.py:24-26) -> z_size = ceil(rna_slc/2); gn^2 * rna_slc <= 512 */
  int32_t n_stain;        /* 2 for stain='all', else 1 */
  int32_t rna_num;        /* 229 */
  int32_t net_ch;         /* model_channels: 64 */
  int32_t ch_mult[4];     /* (1,2,4,8) */
  int32_t embed_ch;       /* embed_channels: 512 */
  int32_t attn_res;       /* attention_resolutions[0]: 16 */
  int32_t num_res_blocks; /* 2 */
  int32_t vis_only;       /* 1: attention-map model (model/unet_attn.py), time_embed + rna_blocks[0] */
  int32_t dtype;          /* TM_DTYPE_F32 | TM_DTYPE_BF16 | TM_DTYPE_F16 (operand type of the convs / Linears / attention) */
} tm_config;

typedef struct tm_model tm_model;

int tm_version(void);
const char* tm_last_error(void);

/* Replaces BeatGANsUNetConfig.make_model() (model/unet_ours.py:78-79,82-275). */
int tm_model_create(const tm_config* cfg, tm_model** out);

/* Replaces model.load_state_dict(state_dct, strict=True) (test_brn.py:140-147), one tensor
 * at a time: `ref_key` is the reference state_dict key (after stripping 'model.'),
 * `host_ptr` a contiguous fp32 HOST tensor of `shape[ndim]`. */
int tm_model_load_param(tm_model* m, const char* ref_key, const void* host_ptr,
                        const int64_t* shape, int ndim, int dtype);

/* strict=True check (every key present), packs / pads / reorders into the device arena
 * (replaces `.to(gpu_id)`, test_brn.py:148).  After this the weights are immutable. */
int tm_model_finalize(tm_model* m);

/* Number of keys the model expects / i-th key (for diagnostics and tests). */
int tm_model_num_params(const tm_model* m);
const char* tm_model_param_key(const tm_model* m, int i);

/* Size of the arena in bytes (after finalize) and raw device pointer: lets the host
 * broadcast rank 0's packed weights over RCCL instead of re-reading the checkpoint on
 * every rank (replaces DDP's construction-time parameter broadcast, test_brn.py:149). */
size_t tm_model_arena_bytes(const tm_model* m);
void* tm_model_arena_ptr(tm_model* m);

/* Workspace (activations, RNA pyramid, skips) needed by tm_unet_forward for
 * `b` images of (p1-1)x(p2-1) interior patches, i.e. b*p1*p2 encoder patches. */
size_t tm_workspace_bytes(const tm_model* m, int b, int p1, int p2, int want_pred2);

/* Replaces BeatGANsUNetModel.forward (model/unet_ours.py:343-426) as called through
 * _WrappedModel.forward (diffusion/diffusion.py:134-154):
 *   x         [b*p1*p2, C, ps, ps] fp32        (C = n_stain*z_size)
 *   t         [b] int64, ORIGINAL-scale timesteps (already mapped by timestep_map)
 *   rna_dense [b*p1*p2, gn, gn, rna_slc*500] fp32 (dense, as test_brn.py:180-181 builds it)
 *   p1, p2    patches per image side including the half-patch padding (= H/ps + 1)
 *   pred      [b*(p1-1)*(p2-1), C, ps, ps]     collage ("o==0") decoder output
 *   pred2     NULL, or [b*p1*p2, C, ps, ps]    original-patch ("o==1") decoder output
 */
int tm_unet_forward(tm_model* m, const void* x, const int64_t* t, const void* rna_dense,
                    int b, int p1, int p2, void* pred, void* pred2_or_null,
                    void* workspace, size_t workspace_bytes, void* stream);

/* The RNA conditioning of a forward (BeatGANsUNetModel.get_rna, model/unet_ours.py:298-323: gene-gene attention block,
 * down_z, the three pyramid convs) depends on the genes only -- not on x or t.  A caller that runs many diffusion steps on
 * the same genes (mode A: LitModel.gen_sample's reverse loop, experiment.py:325-330; the reference recomputes it in every
 * step) computes it ONCE into a buffer of its own and hands it to every step:
 *   tm_rna_pyramid_bytes   size of that buffer for (b, p1, p2)
 *   tm_rna_pyramid         rna_dense [b*p1*p2, gn, gn, rna_slc*500] fp32 -> pyramid (opaque; valid for this model, b, p1, p2)
 *   tm_unet_forward_rna    tm_unet_forward with the precomputed pyramid in place of rna_dense; bit-identical results.
 * The pyramid buffer is only read by tm_unet_forward_rna and may be shared by any number of steps. */
size_t tm_rna_pyramid_bytes(const tm_model* m, int b, int p1, int p2);
int tm_rna_pyramid(tm_model* m, const void* rna_dense, int b, int p1, int p2, void* pyramid, size_t pyramid_bytes,
                   void* stream);
int tm_unet_forward_rna(tm_model* m, const void* x, const int64_t* t, const void* pyramid, size_t pyramid_bytes,
                        int b, int p1, int p2, void* pred, void* pred2_or_null, void* workspace,
                        size_t workspace_bytes, void* stream);

/* The tile sweep (mode B, test_brn.py:232-255) also recomputes the conditioning of the SAME genes at each of its T diffusion
 * steps, but the whole pyramid of a tile (~1 MB per patch) is too large to keep for every tile.  Level 0 -- gene-gene
 * attention -> down_z -> Upsample (model/unet_ours.py:298-310, MBAblocks.py:472-479), the part that reads the gene counts and
 * costs most -- is 59 KB per patch: a caller may keep it per tile / window and skip that part (and the dense gene tensor)
 * from the second step on:
 *   tm_rna_level0_bytes     size of the level-0 tensor for (b, p1, p2) (the activation-stream type of the model: fp32 or 16-bit)
 *   tm_rna_level0           rna_dense -> level0 (16-byte aligned device buffer); workspace >= tm_workspace_bytes(.., 0)
 *   tm_unet_forward_level0  tm_unet_forward with level0 in place of rna_dense; bit-identical results. */
size_t tm_rna_level0_bytes(const tm_model* m, int b, int p1, int p2);
int tm_rna_level0(tm_model* m, const void* rna_dense, int b, int p1, int p2, void* level0, size_t level0_bytes,
                  void* workspace, size_t workspace_bytes, void* stream);
int tm_unet_forward_level0(tm_model* m, const void* x, const int64_t* t, const void* level0, size_t level0_bytes,
                           int b, int p1, int p2, void* pred, void* pred2_or_null, void* workspace,
                           size_t workspace_bytes, void* stream);

/* Per-step scalar coefficients: the float64 tables of GaussianDiffusionBeatGans.__init__
 * (diffusion/base.py:64-109) gathered at index i and cast `.float()` (base.py:643) by
 * the host. */
typedef struct tm_step_coefs {
  float sqrt_recip_alphas_cumprod;     /* base.py:89  */
  float sqrt_recipm1_alphas_cumprod;   /* base.py:90  */
  float posterior_mean_coef1;          /* base.py:100 */
  float posterior_mean_coef2;          /* base.py:103 */
  float sigma;                         /* DDPM: exp(0.5*log_variance) (base.py:479-480), 0 when t==0 */
  float sqrt_alpha_bar_prev;           /* DDIM: sqrt(alpha_bar_prev)      (base.py:492) */
  float sqrt_one_minus_alpha_bar_prev; /* DDIM: sqrt(1 - alpha_bar_prev)  (base.py:493, eta=0) */
} tm_step_coefs;

/* Replaces p_mean_variance's eps re-tiling + x0 / clamp / posterior mean
 * (diffusion/base.py:386-393,423-427), ddm_sample's DDPM / DDIM(eta=0) update (:476-498) and
 * the un-patchify + crop of ddm_sample_loop_progressive (:627-628):
 *   x_patches [b*(P1+1)*(P2+1), C, ps, ps]; eps [b*P1*P2, C, ps, ps] (model pred);
 *   noise     NULL or like x_patches (DDPM);  x_prev_img [b, C, P1*ps, P2*ps]. */
int tm_sampler_step(const tm_step_coefs* coefs, const void* x_patches, const void* eps,
                    const void* noise_or_null, void* x_prev_img, int b, int P1, int P2, int C,
                    int ps, int mode, void* stream);

/* Replaces F.pad(img, halfp) + rearrange(im2tl) (diffusion/base.py:606-607):
 *   img [b, C, P1*ps, P2*ps] -> patches [b*(P1+1)*(P2+1), C, ps, ps], zero border. */
int tm_pad_patchify(const void* img, void* patches, int b, int C, int P1, int P2, int ps,
                    float pad_value, void* stream);

/* Replaces unet_attn.BeatGANsUNetModel.get_rna (model/unet_attn.py:143-173):
 *   rna_dense [B, gn, gn, rna_slc*500] -> attn_out [4, B, G, G] fp32 softmax maps
 *   (3 slice-pair-masked + 1 unmasked), rna_mid [B, G, rna_slc-2, gn, gn]. */
size_t tm_gene_attn_workspace_bytes(const tm_model* m, int B);
int tm_gene_attn(tm_model* m, const void* rna_dense, int B, void* attn_out, void* rna_mid,
                 void* workspace, size_t workspace_bytes, void* stream);

int tm_model_destroy(tm_model* m);

/* ---- tile I/O either side of the path ----------------------------
 * tm_gene_tile_dense replaces MBADataset_tst._getgene + _pad_gn (utils/MBADataset_tst.py:65-91) and the
 * sparse->dense step of BeatGANsUNetModel.get_rna (model/unet_ours.py:301-306) for one gene tile:
 *   crd  int32 [3][nnz] device: (h, w, channel) of the tile's COO transcript counts, h/w in pixels of the
 *        +-128-px padded ROI, channel = slice*500 + gene;   dat fp32 [nnz] device: counts
 *   out  fp32 [gsz][gsz][chan_in + 2*zpad_ch] device: out[h/gblk + shift_h][w/gblk + shift_w][zpad_ch + c] += dat
 *        for every entry whose cell lands inside the gsz x gsz grid; everything else is zero.
 * The reference values are gblk=16, shift = pad/gblk - (roi - roio)/gblk = -6, gsz=20, chan_in=25000,
 * zpad_ch = Z_PAD[rna_slc]*500.  Counts are integers, so the result is independent of the order of the adds. */
int tm_gene_tile_dense(const int32_t* crd, const void* dat, int64_t nnz, int gblk, int shift_h, int shift_w,
                       int gsz, int chan_in, int zpad_ch, void* out, void* stream);

/* Host-side decoder of one Blosc-1 frame (lz4 codec, optional byte shuffle): the chunk encoding zarr 2.14.1 /
 * numcodecs 0.15.0 use by default for the state tiles the reference writes with zarr.save_array
 * (test_brn.py:225) and reads back with zarr.load (utils/MBADataset_tst.py:60, infer_brn.py:76).
 * dst == NULL: only *out_bytes (the decoded size) is set.  No GPU involved. */
int tm_blosc_decompress(const void* src, size_t src_bytes, void* dst, size_t dst_cap, size_t* out_bytes);

/* Measurement hooks (bench.py): while enabled, every launch of the dominant kernel
 * (conv3d_mfma<2,..> in fp32, conv27_bf16 / conv27_f16 in the 16-bit modes: the 3x3x3 implicit-GEMM conv) inside tm_unet_forward is bracketed by two
 * hipEvents recorded on the forward's stream.  tm_profile_collect synchronises on the last
 * event, adds up the bracketed durations and the per-launch work, and resets the counters.
 *   nominal_flops  = 2*Cin*Cout*27*voxels per launch (dense-conv convention; what
 *                    torch.utils.flop_counter counts for the reference's Conv3d)
 *   executed_flops = 2/3 of that for z_size 2 (the always-zero z tap is not issued), 1/3 for z_size 1 (centre
 *                    slice only), all of it for z_size 4 / 8
 *   alg_bytes      = input + packed weights + output bytes, each counted once per launch */
typedef struct tm_prof_stats {
  uint64_t launches;
  double total_ms;
  double nominal_flops;
  double executed_flops;
  double alg_bytes;
} tm_prof_stats;
int tm_profile_enable(tm_model* m, int on);
int tm_profile_collect(tm_model* m, tm_prof_stats* out);

/* ---- single-operator entry points (parity tests of the individual kernels) -------------
 * Layout "CB8": fp32 [N][ceil(C/8)][Z][H][W][8] (channel blocks of 8, zero padded).      */

/* NCDHW fp32 <-> CB8 */
int tm_op_to_cb8(const void* x_ncdhw, void* y_cb8, int N, int C, int Z, int H, int W, void* stream);
int tm_op_from_cb8(const void* x_cb8, void* y_ncdhw, int N, int C, int Z, int H, int W, void* stream);

/* Conv3d on the MFMA implicit-GEMM kernels (replaces nn.Conv3d as used in ResBlock,
 * model/MBAblocks.py:146-148,182-186,220-224, the RNA pyramid convs model/unet_ours.py:290-295
 * and down_z model/MBAblocks.py:472-474).  ksize 1: 1x1x1.  ksize 3 with zmode
 *   0: 3x3x3 pad (1,1,1), Z == 2      1: 1x3x3 pad (0,1,1)      2: 3x3x3 pad (0,1,1) (Zout = Z-2)
 *   3: 3x3x3 pad (1,1,1) of the nearest-x2 UPSAMPLED x (Upsample then Conv3d as in ResBlock(up=True), model/MBAblocks.py:254-258,
 *      blocks.py:362-371), Z == 2, computed on x itself with per-phase 2x2 in-plane weights: y is [N, Cout, Z, 2S, 2S].
 * up2: nearest x2 on (H, W) applied to the output (not with zmode 3).  w [Cout][Cin][kz][3][3] HOST fp32,
 * bias [Cout] HOST fp32; x, y CB8 DEVICE. */
int tm_op_conv_mfma(const void* x_cb8, const void* w_host, const void* bias_host, void* y_cb8,
                    int N, int Cin, int Cout, int Z, int S, int ksize, int zmode, int up2,
                    int tile_variant, void* stream);

/* 16-bit variant of the 3x3x3 pad-1 conv (Z == 2): x fp32 CB8 is rounded to `dtype` (TM_DTYPE_BF16 | TM_DTYPE_F16, RNE) on
 * the device, w rounded on the host; fp32 accumulate, fp32 CB8 output.  waves: 0 = the launcher's choice, 4 | 8 = force
 * the 4-wave (128 x 256 / 64 x 512 tile) or 8-wave (128 x 512 / 64 x 1024) workgroup form.
 * res_h16 (nullable): 16-bit CB8 residual [N][ceil(Cout/8)][2][S][S][8] added in fp32 before the final rounding;
 * y_h16 (nullable): write the result as a 16-bit CB8 tensor of that shape INSTEAD of y_cb8 (the 16-bit activation
 * stream of the model: block outputs and residuals are 16-bit tensors, as under the reference's fp16 autocast).
 * ups != 0: the conv of the nearest-x2 UPSAMPLED x (ResBlock(up=True), model/MBAblocks.py:254-258), computed on x itself with
 * per-phase 2x2 in-plane weights: outputs are [N, Cout, 2, 2S, 2S]; Cout a multiple of 128, no residual.
 * res_half != 0: res_h16 is [N][ceil(Cout/8)][2][S/2][S/2][8] and read at (z, y >> 1, x >> 1) (the residual of that block:
 * the upsampled block input, :297). */
int tm_op_conv27_bf16(const void* x_cb8, const void* w_host, const void* bias_host, void* y_cb8,
                      int N, int Cin, int Cout, int S, int dtype, int waves, const void* res_h16, void* y_h16,
                      int ups, int res_half, void* stream);

/* Timing hook: `iters` launches of that conv in the forms the model uses (16-bit stream output, optional 16-bit
 * residual, fused norm epilogue, upsampled-input form) on random device data in [-1, 1); *ms_per_launch = mean launch
 * duration between two events after one warm-up launch.  waves: 0 | 4 | 8 as above, 9 = the lockstep 8-wave kernel
 * (kept for A/B against the ping-pong form that `8` selects). */
int tm_op_conv27_time(int N, int Cin, int Cout, int S, int dtype, int waves, int ups, int with_res, int fused,
                      int iters, float* ms_per_launch, void* stream);

/* The same conv with the ResBlock mid-section fused into its epilogue (Cout in {64, 128}): out_layers[0]
 * RMSNorm(C) * norm_w -> x * (1 + scale) + shift -> SiLU (model/MBAblocks.py:196-203,356-367), written as the 16-bit CB8
 * tensor a2_out [N][Cout/8][2][S][S][8] (the second conv's input).  norm_w [Cout], scale / shift [ceil(N/per_image)][Cout]
 * HOST fp32; patch n uses row n / per_image. */
int tm_op_conv27_fused(const void* x_cb8, const void* w_host, const void* bias_host, const void* norm_w_host,
                       const void* scale_host, const void* shift_host, void* a2_out, int N, int Cin, int Cout,
                       int S, int per_image, int dtype, int waves, void* stream);

/* 16-bit 1x1x1 conv / Linear over '(z h w) c' tokens (x and w rounded to `dtype`, fp32 accumulate).  waves: 0 | 4 | 8
 * as above (two co-resident 4-wave workgroups per CU, or one 8-wave workgroup).  Epilogue (model/MBAblocks.py:486-489):
 * y = res + gate * gelu?(W x + b) with optional 16-bit CB8 `res_h16` / `gate_h16` and 16-bit output `y_h16`. */
int tm_op_conv1_bf16(const void* x_cb8, const void* w_host, const void* bias_host, void* y_cb8,
                     int N, int Cin, int Cout, int Z, int S, int gelu, int dtype, int waves, const void* res_h16,
                     const void* gate_h16, void* y_h16, void* stream);

/* The ResBlock skip conv as the 16-bit modes run it (model/MBAblocks.py:220-224,297 on x = th.cat((h, skip, rna), 1),
 * model/unet_ours.py:384,418, with to_collage :325-341 applied to the sources of the collage decoder): a 1x1x1 conv whose
 * input is the channel concat of nsrc (1..3) tensors READ IN PLACE -- the concat and the half-patch re-tiling are address
 * rules of the kernel's staging, not tensors.  x_cb8[i]: fp32 CB8 [Ni, cin[i], Z, S, S] (device), Ni = N for a plain
 * source, N / ((p1-1)(p2-1)) * p1 * p2 for a collaged one (collage[i] != 0: the source lives on the (p1 x p2) encoder
 * patch grid of each image and is read at the half-patch-shifted position).  w [Cout][sum cin] HOST.  Output fp32 CB8. */
int tm_op_conv1_concat(const void* const* x_cb8, const int* cin, const int* collage, int nsrc, const void* w_host,
                       const void* bias_host, void* y_cb8, int N, int Cout, int Z, int S, int p1, int p2,
                       int dtype, int waves, void* stream);

/* The block-input pass of the 16-bit modes on its own: th.cat of nsrc (1..3) sources (model/unet_ours.py:384,418) with
 * to_collage (:325-341) where collage[i] != 0, optional resampling (up2 = 1: nearest x2, Upsample, model/blocks.py:362-371,
 * sources at S/2; up2 = 2: the Downsample form of ResBlock(down=True), blocks.py:389-403 -- ONE plain source at 2S, every
 * source voxel normalised and activated with its own statistics, then the 2 x 2 average; raw_h16 = the 2 x 2 average of x), LlamaRMSNorm over the real channel count c_real (model/MBAblocks.py:21-43; norm_w_dev: device fp32 [padded C], or
 * null), modulation (mod 0 none; 1 per image: device fp32 scale / shift rows [b][mod_stride], image = n / per_image,
 * apply_conditions :356-367; 2 per voxel: 16-bit CB8 scale / shift tensors of the output geometry with patch stride
 * mod_stride elements, modulate :608-614), SiLU (act != 0).  Sources and outputs are 16-bit CB8 DEVICE tensors (dtype
 * TM_DTYPE_BF16 / TM_DTYPE_F16); out_h16 has ceil(Cb / 2) * 2 channel blocks (pad blocks zero), raw_h16 (optional) receives
 * the gathered, un-normalised input.  variant: 0 = the form the model picks, 1 = prep_kernel, 2 / 3 = prep_h16_kernel with one
 * wave / four waves per 64 voxels.  iters >= 1 launches; elapsed_ms (host, optional): mean time of launches 2..iters. */
int tm_op_prep_h16(const void* const* src_h16, const int* src_c, const int* collage, int nsrc, int N, int Z, int S,
                   int p1, int p2, int up2, const void* norm_w_dev, int c_real, int mod, const void* mod_scale,
                   const void* mod_shift, long mod_stride, int per_image, int act, int dtype, int variant,
                   void* out_h16, void* raw_h16, int iters, float* elapsed_ms, void* stream);

/* Windowed gene-patch cross attention core (model/MBAblocks.py:551-601 between the q/k/v Linears and proj):
 * q, k, v fp32 CB8 [N, C, Z, S, S]; qw, kw: device fp32 [C] (q_norm / k_norm weights).
 * dtype TM_DTYPE_F32: fp32 MFMA kernels, out = fp32 CB8.  TM_DTYPE_BF16: inputs are rounded to bf16 first (what the
 * bf16 q / kv Linears emit), out = bf16 CB8 [N][C/8][Z][S][S][8]. */
int tm_op_window_attn(const void* q_cb8, const void* k_cb8, const void* v_cb8, const void* qw_dev, const void* kw_dev,
                      void* out, int N, int C, int Z, int S, int dtype, void* stream);

/* Generic direct Conv3d (stem / head / RNA path), NCDHW in, NCDHW out. */
int tm_op_conv_direct(const void* x, const void* w_host, const void* bias_host, void* y, int N,
                      int Cin, int Cout, int Zin, int S, int kz, int ky, int kx, int pz, int py,
                      int px, int silu_in, int up2_out, void* stream);

/* ---- training slice: the pieces of ONE ResBlock's training step ---------------------------------
 * ResBlock._forward in training mode (model/MBAblocks.py:237-299) is
 *   A = SiLU(RMSNorm(x) w1);  H1 = Conv3d(A);  D = Dropout(SiLU(RMSNorm(H1) w2 (1 + scale) + shift));  out = skip(x) + Conv3d(D)
 * and its backward is composed of the entry points below (teramind_amd.training.ResBlockTrain does the composition; the
 * gradient of every piece is checked against torch.autograd of the oracle, tests/test_gpu_train.py).  fp32, CB8 tensors. */

/* y = Dropout(SiLU(RMSNorm_C(x) * norm_w * (1 + scale[img]) + shift[img])) with a SUPPLIED keep mask (fp32 CB8 0/1, or NULL
 * = no dropout; nn.Dropout(p=0.1) draws it in the reference, config_parm.py:46) and drop_scale = 1 / (1 - p).
 * norm_w [C], scale / shift [ceil(N / per_image)][C] HOST (or both NULL: in_layers). */
int tm_op_prep_train(const void* x_cb8, const void* norm_w_host, const void* scale_host, const void* shift_host,
                     const void* mask_cb8, float drop_scale, int per_image, void* y_cb8, int N, int C, int Z, int S,
                     void* stream);

/* Backward of the above: g = dL/dy (CB8) -> dx (CB8), dL/dnorm_w [C], dL/dscale, dL/dshift [nimg][C] (HOST outputs). */
int tm_op_prep_bwd(const void* x_cb8, const void* g_cb8, const void* norm_w_host, const void* scale_host,
                   const void* shift_host, const void* mask_cb8, float drop_scale, int per_image, void* dx_cb8,
                   void* dw_host, void* dscale_host, void* dshift_host, int N, int C, int Z, int S, void* stream);

/* dL/dx of Conv3d(k = 3x3x3 pad 1 (ksize 3) | 1x1x1 (ksize 1)): dy CB8 [N, Cout, Z, S, S] -> dx CB8 [N, Cin, Z, S, S];
 * w [Cout][Cin][k^3] HOST as in the reference state_dict.  Runs on the forward MFMA conv kernel with re-packed weights. */
int tm_op_conv_dgrad(const void* dy_cb8, const void* w_host, void* dx_cb8, int N, int Cin, int Cout, int Z, int S,
                     int ksize, void* stream);

/* dL/dw [Cout][Cin][k^3] and dL/dbias [Cout] (HOST outputs; db may be NULL) of the same convs. */
int tm_op_conv_wgrad(const void* x_cb8, const void* dy_cb8, void* dw_host, void* db_host_or_null, int N, int Cin,
                     int Cout, int Z, int S, int ksize, void* stream);

/* ---- training slice, AttnBlock (model/MBAblocks.py:428-514 AttnBlock.forward, :517-601 Attention, :608-614 modulate) -----
 *   m = adaLN(SiLU(y));  (shift, scale, gate) x (msa, mlp) = m.chunk(6)
 *   x = x + gate_msa * proj(core(q(modulate(norm1(x))), k(y), v(y)));   x = x + gate_mlp * fc2(GELU(fc1(modulate(norm2(x)))))
 * Every Linear is a 1x1x1 conv (tm_op_conv_mfma / tm_op_conv_dgrad / tm_op_conv_wgrad with ksize 1); the pieces below are the
 * rest.  fp32 CB8 device tensors, norm weights HOST.  teramind_amd.training.AttnBlockTrain composes forward and backward;
 * tests/test_gpu_train.py checks every gradient against the reference module's own autograd (tests/golden/train_attn_ref.npz). */

/* Elementwise on n floats: op 0 o1 = a + b * c | 1 o1 = a * b, o2 = a * c | 2 o1 = gelu_tanh(a) | 3 o1 = a * gelu_tanh'(b) |
 * 4 o1 = silu(a) | 5 o1 = a * silu'(b) | 6 o1 = a + b | 7 o1 = 4 a | 8 o1 = a / 4. */
int tm_op_ew(int op, const void* a, const void* b, const void* c, void* o1, void* o2, long n, void* stream);

/* y = RMSNorm_C(x) * norm_w * (1 + scale) + shift with per-voxel scale / shift (CB8 tensors of x's geometry). */
int tm_op_modnorm(const void* x_cb8, const void* norm_w_host, const void* scale_cb8, const void* shift_cb8, void* y_cb8,
                  int N, int C, int Z, int S, void* stream);

/* Backward of the above: g = dL/dy -> dx, dscale, dshift (CB8) and dL/dnorm_w [C] (HOST). */
int tm_op_modnorm_bwd(const void* x_cb8, const void* g_cb8, const void* norm_w_host, const void* scale_cb8, void* dx_cb8,
                      void* dscale_cb8, void* dshift_cb8, void* dw_host, int N, int C, int Z, int S, void* stream);

/* The attention core (q/k RMSNorm, softmax(q k^T / C) v per 2 x 2 window over (h, w) and all z; one head, n_h = 2):
 * dout_cb8 == NULL: forward, writes o_cb8.  dout_cb8 != NULL: backward, writes dq / dk / dv (CB8) and the q/k norm weight
 * gradients [C] (HOST).  Windows of 32, 64 or 128 tokens, C <= 512. */
int tm_op_window_attn_train(const void* q_cb8, const void* k_cb8, const void* v_cb8, const void* qw_host,
                            const void* kw_host, const void* dout_cb8, void* o_cb8, void* dq_cb8, void* dk_cb8,
                            void* dv_cb8, void* dqw_host, void* dkw_host, int N, int C, int Z, int S, void* stream);

/* ---- training slice, the small dense pieces: time embedding (model/unet_ours.py:442-476), ResBlock.emb_layers
 * (model/MBAblocks.py:178-186) and the gene-gene AttnBlock of the RNA pyramid (model/unet_ours.py:277-323) are Linears over
 * [tokens][features] rows.  DEVICE pointers throughout (fp32), deterministic. */

/* C[b](m, n) = alpha * sum_k A[b](m, k) B[b](k, n) (+ bias[n] if bias_mode 1, bias[m] if 2) (+ C if accumulate), every operand
 * addressed through element strides strides9 = {sam, sak, sbk, sbn, scm, scn, sab, sbb, scb} (HOST array). */
int tm_op_gemm_f32(const void* A_dev, const void* B_dev, const void* bias_dev, void* C_dev, int M, int N, int K,
                   const long* strides9_host, int batch, int bias_mode, int accumulate, float alpha, void* stream);

/* Row-wise ops on [rows][D]: 0 y = RMSNorm_D(x) * w | 1 y = dL/dx of op 0 given g (and dw_dev[D] = dL/dw) | 2 y = softmax_D(x) |
 * 3 y = x * (g - sum_D(g x)) (softmax backward, x = the saved probabilities). */
int tm_op_rows(int op, const void* x_dev, const void* w_dev, const void* g_dev, void* y_dev, void* dw_dev, long rows, int D,
               void* stream);

/* (h, w) resampling of a CB8 tensor: mode 1 nearest x2 (source at S_out / 2), mode 2 AvgPool(1,2,2) (source at 2 S_out)
 * (model/blocks.py:362-403). */
int tm_op_resample(const void* x_cb8, void* y_cb8, int N, int C, int Z, int S_out, int mode, void* stream);

/* ---- training slice, the optimizer step (experiment.py:207-219 clip_grad_norm_, :394-414 torch.optim.Adam) on one flat fp32
 * device arena of parameters / gradients / first and second moments. */

/* *out_host = sum of x[i]^2 (two-stage, fixed order). */
int tm_op_sumsq(const void* x_dev, long n, float* out_host, void* stream);

/* One torch.optim.Adam step (amsgrad off): g' = g * grad_scale + weight_decay * p; m, v, p updated in place; step >= 1. */
int tm_op_adam(void* p_dev, const void* g_dev, void* m_dev, void* v_dev, long n, float lr, float beta1, float beta2,
               float eps, float weight_decay, int step, float grad_scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TERAMIND_HIP_H */
