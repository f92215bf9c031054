"""Shared helpers of the parity tests (HIP path vs oracle/)."""
import ctypes as C
import os

import numpy as np
import torch

from teramind_amd import _lib
from teramind_amd.config import PathConfig
from teramind_amd.weights import hashed_state_dict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

_SD = {}


def state_dict(cfg: PathConfig = None, seed=0, vis_only=False):
    cfg = cfg or PathConfig()
    key = (cfg.name, seed, vis_only)
    if key not in _SD:
        _SD[key] = hashed_state_dict(cfg, seed, vis_only)
    return _SD[key]


def rand_int(shape, lo, hi, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi + 1, shape, generator=g).float()


def to_cb8(x):
    """NCDHW cuda tensor -> CB8 cuda tensor via the library."""
    N, Cc, Z, H, W = x.shape
    cb = (Cc + 7) // 8
    y = torch.empty((N, cb, Z, H, W, 8), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().tm_op_to_cb8(_lib.ptr(x.contiguous()), _lib.ptr(y), N, Cc, Z, H, W, _lib.current_stream_ptr()))
    return y


def from_cb8(y, Cc):
    N, cb, Z, H, W, _ = y.shape
    x = torch.empty((N, Cc, Z, H, W), dtype=torch.float32, device=y.device)
    _lib.check(_lib.lib().tm_op_from_cb8(_lib.ptr(y), _lib.ptr(x), N, Cc, Z, H, W, _lib.current_stream_ptr()))
    return x


def conv_mfma(x, w, b, ksize, variant=0, zmode=0, up2=False):
    """x NCDHW cuda; w, b host tensors.  Returns NCDHW cuda output of the MFMA conv."""
    N, Cin, Z, S, _ = x.shape
    Cout = w.shape[0]
    Zo = Z - 2 if (ksize == 3 and zmode == 2) else Z
    So = 2 * S if (up2 or zmode == 3) else S
    xc = to_cb8(x)
    yc = torch.zeros((N, (Cout + 7) // 8, Zo, So, So, 8), dtype=torch.float32, device=x.device)
    wh, bh = w.contiguous().float(), b.contiguous().float()
    _lib.check(_lib.lib().tm_op_conv_mfma(_lib.ptr(xc), C.c_void_p(wh.data_ptr()), C.c_void_p(bh.data_ptr()), _lib.ptr(yc),
                                          N, Cin, Cout, Z, S, ksize, zmode, int(up2), variant, _lib.current_stream_ptr()),
               "tm_op_conv_mfma")
    return from_cb8(yc, Cout), yc


H16 = {"bf16": (1, torch.bfloat16), "f16": (2, torch.float16)}


def to_cb8_h16(x, dtype):
    """NCDHW fp32 cuda -> 16-bit CB8 cuda tensor [N][ceil(C/8)][Z][H][W][8]."""
    return to_cb8(x).to(H16[dtype][1])


def conv27_bf16(x, w, b, dtype="bf16", waves=0, res=None, out16=False, ups=False, res_half=False):
    """x NCDHW cuda (Z == 2); rounds x (device) and w (host) to the 16-bit `dtype`, fp32 accumulate.
    waves: 0 = the launcher's choice, 4 / 8 = force that workgroup form.  res: NCDHW residual handed over as a 16-bit CB8
    tensor; out16: take the result as a 16-bit CB8 tensor (the model's activation-stream form).  Returns fp32 NCDHW."""
    N, Cin, Z, S, _ = x.shape
    Cout = w.shape[0]
    xc = to_cb8(x)
    So = 2 * S if ups else S       # ups: the conv of the nearest-x2 upsampled x, computed on x; res_half: res lives at So / 2
    shp = (N, (Cout + 7) // 8, Z, So, So, 8)
    yc = torch.zeros(shp, dtype=torch.float32, device=x.device) if not out16 else None
    yh = torch.zeros(shp, dtype=H16[dtype][1], device=x.device) if out16 else None
    rh = to_cb8_h16(res, dtype) if res is not None else None
    wh, bh = w.contiguous().float(), b.contiguous().float()
    _lib.check(_lib.lib().tm_op_conv27_bf16(_lib.ptr(xc), C.c_void_p(wh.data_ptr()), C.c_void_p(bh.data_ptr()), _lib.ptr(yc),
                                            N, Cin, Cout, S, H16[dtype][0], waves, _lib.ptr(rh), _lib.ptr(yh),
                                            int(ups), int(res_half), _lib.current_stream_ptr()), "tm_op_conv27_bf16")
    yc = yh.float() if out16 else yc
    return from_cb8(yc, Cout), yc


def conv27_fused(x, w, b, norm_w, scale, shift, per_image, dtype="bf16", waves=0):
    """3x3x3 conv + fused RMSNorm(C) * norm_w -> (1 + scale) + shift -> SiLU epilogue; returns the 16-bit result as
    fp32 NCDHW.  scale / shift: [ceil(N / per_image), Cout] host tensors."""
    N, Cin, Z, S, _ = x.shape
    Cout = w.shape[0]
    xc = to_cb8(x)
    a2 = torch.zeros((N, Cout // 8, Z, S, S, 8), dtype=H16[dtype][1], device=x.device)
    hs = [t.contiguous().float() for t in (w, b, norm_w, scale, shift)]
    _lib.check(_lib.lib().tm_op_conv27_fused(_lib.ptr(xc), *[C.c_void_p(t.data_ptr()) for t in hs], _lib.ptr(a2),
                                             N, Cin, Cout, S, per_image, H16[dtype][0], waves, _lib.current_stream_ptr()),
               "tm_op_conv27_fused")
    return from_cb8(a2.float(), Cout)


def conv1_bf16(x, w, b, gelu=False, dtype="bf16", waves=0, res=None, gate=None, out16=False):
    N, Cin, Z, S, _ = x.shape
    Cout = w.shape[0]
    xc = to_cb8(x)
    shp = (N, (Cout + 7) // 8, Z, S, S, 8)
    yc = torch.zeros(shp, dtype=torch.float32, device=x.device) if not out16 else None
    yh = torch.zeros(shp, dtype=H16[dtype][1], device=x.device) if out16 else None
    rh = to_cb8_h16(res, dtype) if res is not None else None
    gh = to_cb8_h16(gate, dtype) if gate is not None else None
    wh, bh = w.contiguous().float(), b.contiguous().float()
    _lib.check(_lib.lib().tm_op_conv1_bf16(_lib.ptr(xc), C.c_void_p(wh.data_ptr()), C.c_void_p(bh.data_ptr()), _lib.ptr(yc),
                                           N, Cin, Cout, Z, S, int(gelu), H16[dtype][0], waves, _lib.ptr(rh), _lib.ptr(gh),
                                           _lib.ptr(yh), _lib.current_stream_ptr()), "tm_op_conv1_bf16")
    yc = yh.float() if out16 else yc
    return from_cb8(yc, Cout), yc


def conv_direct(x, w, b, pad, silu_in=False, up2=False):
    N, Cin, Zin, S, _ = x.shape
    Cout, _, kz, ky, kx = w.shape
    Zout = Zin + 2 * pad[0] - kz + 1
    So = 2 * S if up2 else S
    y = torch.empty((N, Cout, Zout, So, So), dtype=torch.float32, device=x.device)
    wh, bh = w.contiguous().float(), b.contiguous().float()
    _lib.check(_lib.lib().tm_op_conv_direct(_lib.ptr(x.contiguous()), C.c_void_p(wh.data_ptr()), C.c_void_p(bh.data_ptr()),
                                            _lib.ptr(y), N, Cin, Cout, Zin, S, kz, ky, kx, pad[0], pad[1], pad[2],
                                            int(silu_in), int(up2), _lib.current_stream_ptr()), "tm_op_conv_direct")
    return y


def report(name, got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    d = (got - ref).abs()
    return (f"{name}: max|d|={d.max().item():.3e} mean|d|={d.mean().item():.3e} "
            f"ref absmax={ref.abs().max().item():.3e} std={ref.std().item():.3e} "
            f"nan={int(torch.isnan(got).sum())}")


def synthetic_gene_coo(row, col, total_slc=50, nnz=20000, seed=0, size=256, dtype=np.uint16):
    """A gene tile the way the reference stores it: COO counts over the tile padded by size/2 px,
    shape [2*size, 2*size, total_slc*500]; duplicates and entries in the cropped margin included."""
    rng = np.random.default_rng(seed * 1000003 + row * 1009 + col)
    shape = (2 * size, 2 * size, total_slc * 500)
    crd = np.stack([rng.integers(0, shape[0], nnz), rng.integers(0, shape[1], nnz), rng.integers(0, shape[2], nnz)])
    crd[:, : nnz // 10] = crd[:, nnz // 10: 2 * (nnz // 10)]          # repeated coordinates must add up
    data = rng.integers(1, 6, nnz).astype(dtype)
    return data, crd.astype(np.int64), shape


def write_gene_dir(gdir, rows, cols, total_slc=50, nnz=20000, seed=0):
    from teramind_amd import formats, tiles
    os.makedirs(gdir, exist_ok=True)
    for r in rows:
        for c in cols:
            half = tiles.TILE // 2
            v = (r * 256, r * 256 + 256, c * 256, c * 256 + 256, r * 256 - half, r * 256 + 256 + half, c * 256 - half, c * 256 + 256 + half)
            data, crd, shape = synthetic_gene_coo(r, c, total_slc, nnz, seed)
            formats.write_gene_npz(os.path.join(gdir, "_".join(map(str, v)) + ".npz"), data, crd, shape)


def prep_h16(xs, cins, flags, b, p1, p2, S, up2=False, norm_w=None, mod=0, scale=None, shift=None, per_image=1, act=True,
             dtype="bf16", variant=0, want_raw=False, iters=1):
    """The 16-bit block-input pass (tm_op_prep_h16).  xs: NCDHW fp32 sources (cuda), handed over as 16-bit CB8 tensors;
    norm_w: list of per-source fp32 weight vectors (or None); mod 1: scale / shift [images][C] fp32, mod 2: NCDHW tensors of
    the output geometry (handed over 16-bit).  Returns (list of per-source fp32 NCDHW slices of the output, the same for the
    raw copy or None, mean ms per launch)."""
    dev = xs[0].device
    td = H16[dtype][1]
    any_col = any(flags)
    N = b * (p1 - 1) * (p2 - 1) if any_col else xs[0].shape[0]
    Z = xs[0].shape[2]
    xc = [to_cb8_h16(x, dtype) for x in xs]
    cbs = [(c + 7) // 8 for c in cins]
    cbtot = sum(cbs)
    cbe = (cbtot + 1) // 2 * 2
    nw = None
    if norm_w is not None:
        nw = torch.zeros(cbtot * 8, dtype=torch.float32, device=dev)
        o = 0
        for w, c, cb in zip(norm_w, cins, cbs):
            nw[o:o + c] = w.to(dev)
            o += cb * 8
    sc = sh = None
    stride = 0
    if mod == 1:
        sc, sh = scale.contiguous().float().to(dev), shift.contiguous().float().to(dev)
        stride = sc.shape[1]
    elif mod == 2:
        sc, sh = to_cb8_h16(scale.to(dev), dtype), to_cb8_h16(shift.to(dev), dtype)
        stride = sc[0].numel()
    out = torch.full((N, cbe, Z, S, S, 8), 7.0, dtype=td, device=dev)
    raw = torch.full((N, cbe, Z, S, S, 8), 7.0, dtype=td, device=dev) if want_raw else None
    ptrs = (C.c_void_p * len(xc))(*[t.data_ptr() for t in xc])
    cin = (C.c_int * len(xc))(*cins)
    col = (C.c_int * len(xc))(*[int(f) for f in flags])
    ms = C.c_float(0.0)
    _lib.check(_lib.lib().tm_op_prep_h16(ptrs, cin, col, len(xc), N, Z, S, p1, p2, int(up2), _lib.ptr(nw), sum(cins), mod,   # up2: 0 | 1 | 2 (down)
                                         _lib.ptr(sc), _lib.ptr(sh), stride, per_image, int(act), H16[dtype][0], variant,
                                         _lib.ptr(out), _lib.ptr(raw), iters, C.byref(ms), _lib.current_stream_ptr()),
               "tm_op_prep_h16")

    def split(t):
        res, o = [], 0
        for c, cb in zip(cins, cbs):
            res.append(from_cb8(t[:, o:o + cb].float().contiguous(), c))
            o += cb
        pad = t[:, cbtot:]
        assert not pad.numel() or float(pad.float().abs().max()) == 0.0, "pad blocks must be zero"
        return res
    return split(out), (split(raw) if want_raw else None), ms.value
