// Internal launcher interface between the executor (tm_model.hip) and the kernels
// (tm_kernels.hip).  Not part of the public ABI (include/teramind_hip.h is).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tmk {

// CB8 activation tensor view: fp32 [N][Cb][Z][H][W][8]; channel c lives in block c/8,
// slot c%8; padded slots are kept at exactly 0.
struct TV {
  float* p = nullptr;
  int N = 0, C = 0, Cb = 0, Z = 0, H = 0, W = 0;
  long nstride = 0;                 // floats between consecutive n (>= Cb*Z*H*W*8)
  long plane() const { return (long)Z * H * W * 8; }   // floats per channel block
  // view of channel blocks [cb0, cb0+ncb) (C is set to ncb*8)
  TV blocks(int cb0, int ncb) const {
    TV v = *this;
    v.p = p + (long)cb0 * plane();
    v.Cb = ncb;
    v.C = ncb * 8;
    return v;
  }
};

// ---- packed conv weights for the MFMA implicit-GEMM kernel ---------------------------
// layout: [n_tile][cblk][tap][64 cout][8 cin] fp32, cout padded to a multiple of 64,
// cin padded per concat segment to multiples of 8 ("virtual" cin order).
struct ConvW {
  const float* w = nullptr;     // device
  const float* bias = nullptr;  // device, [ntile*64]
  int Cout = 0, Cbi = 0, taps = 0, ntile = 0;
};
size_t conv_pack_floats(int Cout, int Cbi, int taps);
// host-side packing; seg_c[i] = real channels of concat segment i (each padded to x8).
void conv_pack_host(const float* w /*[Cout][Cin][taps]*/, int Cout, const int* seg_c, int nseg,
                    int taps, float* out);
size_t conv_pack_ups_floats(int Cout, int Cbi);          // phase weights of the upsampled-input conv (ZM_UPS), taps = 12
void conv_pack_ups_host(const float* w /*[Cout][Cin][27]*/, int Cout, const int* seg_c, int nseg, float* out);
void vec_pack_host(const float* v, const int* seg_c, int nseg, float* out);  // per-cin vector -> virtual order

enum { EPI_NONE = 0, EPI_GELU = 1, EPI_UP2 = 2 };
enum { ZM_PAD1 = 0, ZM_INPLANE = 1, ZM_VALID = 2,    // z structure of a k x 3 x 3 conv (see conv3d_mfma)
       ZM_UPS = 3 };                                   // 3x3x3 pad 1 of nearest-x2 upsampled x, computed on x (conv_pack_ups_host weights)
struct ConvLaunch {
  TV x;                  // activated input, Cb == w.Cbi
  ConvW w;
  TV y;                  // output (for EPI_UP2: H,W = 2*x.H)
  const TV* res = nullptr;   // optional residual (same geometry as y)
  const TV* gate = nullptr;  // optional gate: y = res + gate * (conv + bias)
  int flags = 0;
  int tile_variant = 0;  // 0 auto, 1 = 128-voxel blocks, 2 = 256-voxel blocks
  int gate_half = 0;     // taps == 1 only: `gate` lives at S/2 and is read at (z, y >> 1, x >> 1) (S a power of two)
  int res_half = 0;      // k x 3 x 3 only: `res` lives at S/2 and is read at (z, y >> 1, x >> 1): the residual of a ResBlock(up=True)
                         // is the nearest-x2 upsampled block input (model/MBAblocks.py:254-258,297)
  int zmode = ZM_PAD1;   // ignored for taps == 1
};
hipError_t launch_conv_mfma(const ConvLaunch& L, hipStream_t s);

// ---- bf16 3x3x3 conv (tm_conv_bf16.hip) --------------------------------------------------
struct TVH {                        // bf16 CB8 tensor [N][Cb (even)][Z][H][W][8]; strides in elements
  uint16_t* p = nullptr;
  int N = 0, C = 0, Cb = 0, Z = 0, H = 0, W = 0;
  long nstride = 0;
  TVH blocks(int cb0, int ncb) const {          // channel-block slice sharing the storage
    TVH t = *this;
    t.p = p + (long)cb0 * Z * H * W * 8;
    t.Cb = ncb; t.C = ncb * 8;
    return t;
  }
};
struct ConvLaunchH {
  TVH x;
  const uint16_t* w = nullptr;      // packed by conv_bf16_pack_host
  const float* bias = nullptr;      // [ceil(Cout/64)*64] fp32
  int Cout = 0;
  TV y;                             // fp32 CB8 output (geometry is also used for the bf16 output)
  const TV* res = nullptr;
  const TV* gate = nullptr;         // conv1 only
  const TVH* gate_h = nullptr;      // conv1 only: bf16 gate instead of `gate`
  int flags = 0;                    // conv1 only: EPI_GELU
  uint16_t* y_h = nullptr;          // write 16-bit CB8 INSTEAD of y (the 16-bit activation stream / next Linear's input)
  long yh_nstride = 0;
  const TVH* res_h = nullptr;       // 16-bit CB8 residual instead of `res`
  // conv27 only, Cout in {64, 128}: fuse RMSNorm(C)*w -> x(1+scale)+shift -> SiLU into the epilogue and write
  // the bf16 tensor `a2` (the next conv's input) instead of y
  int fuse_norm = 0;
  const float *norm_w = nullptr, *mod_scale = nullptr, *mod_shift = nullptr;
  long mod_stride = 0;
  int per_image = 1;
  TVH a2;
  int force_waves = 0;              // 0 = auto, 4 | 8 = workgroup form (tests / A-B; TM_CONV27_WAVES, TM_CONV1_WAVES)
  int gate_half = 0;                // conv1 only: gate_h lives at S/2 and is read at (z, y >> 1, x >> 1) (S a power of two)
  int res_half = 0;                 // conv27 only: res_h lives at S/2 and is read at (z, y >> 1, x >> 1)
  int ups = 0;                      // conv27 only: x is the LOW-resolution tensor (y.H == 2 x.H), w = conv_bf16_pack_ups_host weights:
                                    // the conv of the nearest-x2 upsampled x (Cout a multiple of 128, Z == 2, no residual)
  // conv1 only: input = channel concat of nsrc (1..3) 16-bit CB8 tensors read in place, each optionally through the collage
  // remap of a (p1 x p2) source patch grid; `x` then only carries N, Z, H, W and the even-padded block count (x.p unused)
  int nsrc = 0;
  TVH xs[3];
  int xs_collage[3] = {0, 0, 0};
  int p1 = 0, p2 = 0;
};
hipError_t init_bf16_device();      // per-device set-up that must not happen lazily inside a stream capture
hipError_t init_f16_device();
int conv_bf16_tn(int Cout);
size_t conv1_bf16_pack_elems(int Cout, int Cbi);
void conv1_bf16_pack_host(const float* w, int Cout, const int* seg_c, int nseg, uint16_t* out);
hipError_t launch_conv1_bf16(const ConvLaunchH& L, hipStream_t s);
size_t conv_bf16_pack_elems(int Cout, int Cbi);
void conv_bf16_pack_host(const float* w, int Cout, const int* seg_c, int nseg, uint16_t* out);
size_t conv_bf16_pack_ups_elems(int Cout, int Cbi);
void conv_bf16_pack_ups_host(const float* w, int Cout, const int* seg_c, int nseg, uint16_t* out);
void conv_f16_pack_ups_host(const float* w, int Cout, const int* seg_c, int nseg, uint16_t* out);
hipError_t launch_conv27_bf16(const ConvLaunchH& L, hipStream_t s);
hipError_t launch_window_attn_bf16(const TVH& q, const TVH& k, const TVH& v, const float* qnorm_w, const float* knorm_w,
                                   TVH o, hipStream_t s);
// IEEE-half twins of the five functions above (tm_conv_bf16.hip built with -DTM_H16_F16): same layouts, fp16 elements
void conv1_f16_pack_host(const float* w, int Cout, const int* seg_c, int nseg, uint16_t* out);
hipError_t launch_conv1_f16(const ConvLaunchH& L, hipStream_t s);
void conv_f16_pack_host(const float* w, int Cout, const int* seg_c, int nseg, uint16_t* out);
hipError_t launch_conv27_f16(const ConvLaunchH& L, hipStream_t s);
hipError_t launch_window_attn_f16(const TVH& q, const TVH& k, const TVH& v, const float* qnorm_w, const float* knorm_w,
                                  TVH o, hipStream_t s);

// ---- prep: concat + resample + RMSNorm(C) * w -> modulate -> act ---------------------
struct PrepSrc {
  const float* p = nullptr;
  long nstride = 0;
  int Cb = 0;
  int collage = 0;
};
enum { RS_SAME = 0, RS_UP2 = 1, RS_DOWN2 = 2,
       RS_PICK2 = 3 };   // output (z, y, x) at S <- source (z, 2y, 2x) at 2S (then the collage remap, if any): the distinct values of
                         // a nearest-x2 upsampled tensor
enum { MOD_NONE = 0, MOD_IMAGE = 1, MOD_VOXEL = 2 };
struct PrepLaunch {
  PrepSrc src[3];
  int nsrc = 1;
  int resample = RS_SAME;
  int N = 0, Z = 0, S = 0;          // OUTPUT patches / plane size
  int p1 = 0, p2 = 0;               // source patch grid per image (collage only)
  const float* norm_w = nullptr;    // [Cbtot*8] virtual order, or null (no norm)
  float inv_c = 0.f;                // 1 / real channel count
  int mod = MOD_NONE;
  const float* mod_scale = nullptr; // MOD_IMAGE: [b][..] row stride mod_stride; MOD_VOXEL: CB8 tensor
  const float* mod_shift = nullptr;
  long mod_stride = 0;              // MOD_IMAGE: floats per image row; MOD_VOXEL: nstride
  const uint16_t* mod_scale_h = nullptr;   // MOD_VOXEL with a bf16 CB8 modulation tensor (instead of mod_scale/shift)
  const uint16_t* mod_shift_h = nullptr;
  int mod_half = 0;                 // MOD_VOXEL: the modulation tensors live at S/2 and are read at (z, y >> 1, x >> 1)
  int per_image = 1;                // output patches per image (n -> image index)
  int act = 0;                      // 1 = SiLU
  float* out = nullptr;
  long out_nstride = 0;
  float* raw = nullptr;             // optional un-normalised (resampled, concatenated) copy
  long raw_nstride = 0;
  uint16_t* out_h = nullptr;        // bf16 output instead of `out` (conv27_bf16 input), nstride in elements
  long out_h_nstride = 0;
  int pad_blocks = 0;               // extra all-zero channel blocks appended to the bf16 output (pair padding)
  uint16_t* raw_h = nullptr;        // bf16 instead of `raw` (input of the bf16 skip conv); same pad_blocks
  long raw_h_nstride = 0;
  int h_f16 = 0;                    // the 16-bit tensors (out_h, raw_h, mod_*_h, sources with src_h) are IEEE half instead of bf16
  int src_h = 0;                    // the SOURCES are 16-bit CB8 tensors (src[k].p reinterpreted; nstride in elements)
  // training forward (nn.Dropout(p) of ResBlock.out_layers, model/MBAblocks.py:196-203) with a SUPPLIED keep mask:
  // out = act(...) * drop_mask * drop_scale; drop_mask is an fp32 CB8 tensor of the output geometry (0 / 1) or null
  const float* drop_mask = nullptr;
  long drop_ns = 0;
  float drop_scale = 1.f;
};
hipError_t launch_prep(const PrepLaunch& L, hipStream_t s);
// test / measurement knob (tm_op_prep_h16): 0 automatic, 1 prep_kernel, 2 / 3 the two forms of prep_h16_kernel; process-wide,
// set and reset around the op's own launches only
void set_prep_variant(int v);

// ---- generic direct conv (VALU) with strided accessors --------------------------------
struct Acc5 {                        // address = n*sN + (c/8)*sCb + (c%8)*sC8 + z*sZ + y*sY + x*sX
  long sN = 0, sCb = 0, sC8 = 0, sZ = 0, sY = 0, sX = 0;
};
Acc5 acc_ncdhw(int C, int Z, int H, int W);
Acc5 acc_cb8(const TV& t);
struct DirectLaunch {
  const float* x = nullptr; Acc5 ax;
  float* y = nullptr; Acc5 ay;
  const float* w = nullptr;          // device [Cout][Cin][kz][ky][kx]
  const float* bias = nullptr;       // device [Cout]
  int N = 0, Cin = 0, Cout = 0;
  int Zin = 0, Zout = 0, S = 0;      // in-plane size S x S (same in/out)
  int kz = 1, ky = 1, kx = 1, pz = 0, py = 0, px = 0;
  int silu_in = 0, up2_out = 0;
};
hipError_t launch_conv_direct(const DirectLaunch& L, hipStream_t s);

// stem: x NCHW '(s z) h w' [N][Cin*Z][S][S] -> y CB8; w [9][Cin][y.Cb*8]
// y_h != null: write the 16-bit CB8 tensor (geometry of y, nstride in elements) instead of y
hipError_t launch_stem(const float* x, TV y, const float* w, const float* bias, int Cin, hipStream_t s,
                       uint16_t* y_h = nullptr, long yh_nstride = 0, int h_f16 = 0);
// head: x CB8 (Cb blocks) -> y NCHW [N][Cout*Z][S][S]; w [9][x.Cb*8][8]
// h16: 0 = x is fp32 CB8; 1 / 2 = x.p points at a bf16 / fp16 CB8 tensor of the same geometry (nstride in elements)
hipError_t launch_head(TV x, float* y, const float* w, const float* bias, int Cout, hipStream_t s, int h16 = 0);

// ---- layout converters ----------------------------------------------------------------
hipError_t launch_to_cb8(const float* x, TV y, hipStream_t s);        // NCDHW -> CB8 (zero pads)
hipError_t launch_from_cb8(TV x, float* y, hipStream_t s);            // CB8 -> NCDHW

// ---- time embedding --------------------------------------------------------------------
// te[b][E] = W2 * silu(W1 * sinusoid(t) + b1) + b2 ; then ss[b][tot] = Wall * silu(te) + ball
hipError_t launch_time_embed(const int64_t* t, int b, int ch, int E, const float* w1,
                             const float* b1, const float* w2, const float* b2, float* te,
                             hipStream_t s);
hipError_t launch_emb_all(const float* te, int b, int E, const float* wall, const float* ball,
                          int tot, float* ss, hipStream_t s);

// ---- gene-gene attention ---------------------------------------------------------------
struct GeneW {       // all device pointers; matrices stored TRANSPOSED [in][out]
  const float *wq_t, *bq, *wv_t, *bv, *qnorm, *wp_t, *bp, *norm2, *w1_t, *b1, *w2_t, *b2;
};
// rna dense [B][gn][gn][zs*500] -> tokens out as CB8 [B][ceil(G/8)][zs][gn][gn][8] (pad slots untouched);
// zmask_lo/hi: slices outside [lo,hi) are treated as zero (attention-map variants).
hipError_t launch_gene_attn(const float* rna, int B, int gn, int zs, int G, const GeneW& w,
                            float* out_tok /*nullable*/, float* attn_map /*nullable [B][G][G]*/,
                            int zmask_lo, int zmask_hi, hipStream_t s);
// generic G <= 512, D <= 512 form (global scratch `ws`: B * gene_generic_split(B) * gene_generic_ws_floats(G, D) floats;
// gidx: gene slot table or null)
int gene_generic_split(int B);
size_t gene_generic_ws_floats(int G, int D);
hipError_t launch_gene_attn_generic(const float* rna, int B, int gn, int zs, int G, int D, const GeneW& w, const int* gidx,
                                    float* out_tok, float* attn_map, int zmask_lo, int zmask_hi, float* ws, hipStream_t s);
hipError_t launch_rna_mid(const float* rna, int B, int gn, int zs, int G, float* out, hipStream_t s);

// ---- windowed cross attention core -----------------------------------------------------
// q, k, v: CB8 token tensors (tokens = voxels (z h w)); n_h x n_h windows over (H, W).
// o_h != null: write the result as bf16 CB8 (nstride in elements) instead of into `o`.
hipError_t launch_window_attn(const TV& q, const TV& k, const TV& v, const float* qnorm_w,
                              const float* knorm_w, TV o, hipStream_t s, uint16_t* o_h = nullptr,
                              long o_h_nstride = 0);

// ---- sampler ---------------------------------------------------------------------------
struct StepCoefs { float c_recip, c_recipm1, pm1, pm2, sigma, sab_prev, s1m_ab_prev; };
hipError_t launch_sampler_step(const StepCoefs& c, const float* x_patches, const float* eps,
                               const float* noise, float* out, int b, int P1, int P2, int C, int ps,
                               int mode, hipStream_t s);
hipError_t launch_pad_patchify(const float* img, float* patches, int b, int C, int P1, int P2,
                               int ps, float pad, hipStream_t s);

// ---- training slice (tm_train.hip): backward of the prep chain, conv weight gradient, bias gradient ----
hipError_t launch_prep_bwd(const float* x, long x_ns, const float* g, long g_ns, const float* mask, long mask_ns, float drop_scale,
                           const float* w, const float* scale, const float* shift, long mod_stride, int per_image, float* dx,
                           long dx_ns, float* dw, float* dscale, float* dshift, int N, int Cb, int C_real, int Z, int S,
                           float* scratch, hipStream_t s);           // scratch: prep_bwd_scratch_floats(...) floats (two-stage sums)
size_t prep_bwd_scratch_floats(int N, int Cb, int Z, int S, bool with_mod);
hipError_t launch_conv_wgrad(const TV& x, const TV& dy, float* dw, int Cin, int Cout, int taps, hipStream_t s);
hipError_t launch_chan_sum(const TV& x, float* out, int C, hipStream_t s);
// AttnBlock pieces (model/MBAblocks.py:428-614)
hipError_t launch_ew(int op, const float* a, const float* b, const float* c, float* o1, float* o2, long n, hipStream_t s);
hipError_t launch_modnorm_bwd(const TV& x, const float* g, const float* w, const float* scale, float* dx, float* dscale, float* dshift,
                              float* dw, int C_real, float* scratch, hipStream_t s);   // scratch: ceil(voxels / 64) * Cb * 8 floats
hipError_t launch_attn_train(const TV& q, const TV& k, const TV& v, const float* qw, const float* kw, const float* dout, float* o,
                             float* dq, float* dk, float* dv, float* dqw, float* dkw, float* scratch, bool bwd,
                             hipStream_t s);                                            // scratch (bwd): 2 * N * 4 * Cb * 8 floats

hipError_t launch_gemm_f32(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, const long* strides9, int batch,
                           int bias_mode, int accumulate, float alpha, hipStream_t s);   // strides: sam sak sbk sbn scm scn sab sbb scb
hipError_t launch_rows(int op, const float* x, const float* w, const float* g, float* y, float* dw, float* scratch, long rows, int D,
                       hipStream_t s);                                                   // scratch (op 1): ceil(rows / 4) * D floats
hipError_t launch_sumsq(const float* x, long n, float* out, float* scratch, int nwg, hipStream_t s);      // scratch: nwg floats
hipError_t launch_adam(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, float wd, int step,
                       float gscale, hipStream_t s);

// ---- tile I/O (tm_io.hip) --------------------------------------------------------------
int io_fail(int code, const char* msg);     // sets the tm_last_error() text, returns code
hipError_t launch_gene_tile_scatter(const int32_t* crd, const float* dat, long nnz, int gblk, int shift_h, int shift_w,
                                    int gsz, int chan_in, int zpad_ch, float* out, hipStream_t s);
int blosc_decompress(const void* src, size_t src_bytes, void* dst, size_t dst_cap, size_t* out_bytes);

}  // namespace tmk
