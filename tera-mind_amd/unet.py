"""Host-side mirror of the reference model objects for this path.

`BeatGANsUNetModel` keeps the call surface of the reference class of the same name
(reference model/unet_ours.py:82-426): `model(x=, t=, rna=, imgs=, patch_size=, idx=, pos=, ...)`
returning `AutoencReturn(pred, pred2, cond)`, `load_state_dict(sd, strict=True)` with the
reference's key names.  All arithmetic happens in libteramind_hip.so (tm_unet_forward);
torch is used for device memory and streams only.

`GeneAttnModel` mirrors the attention-map model (reference model/unet_attn.py:143-217).
"""
import ctypes as C
from typing import NamedTuple, Optional

import torch

from . import _lib
from .config import PathConfig
from .weights import param_spec


class AutoencReturn(NamedTuple):          # reference model/unet_ours.py:429-432
    pred: torch.Tensor
    pred2: Optional[torch.Tensor]
    cond: Optional[torch.Tensor] = None


def _tm_config(cfg: PathConfig, vis_only: bool) -> _lib.TmConfig:
    if len(cfg.ch_mult) != 4 or len(cfg.attn_res) != 1:
        raise NotImplementedError("ch_mult must have 4 levels and attn_res one resolution")
    c = _lib.TmConfig()
    c.patch_size, c.rna_slc, c.n_stain, c.rna_num = cfg.patch_size, cfg.rna_slc, cfg.n_stain, cfg.rna_num
    c.net_ch, c.embed_ch, c.attn_res = cfg.net_ch, cfg.embed_ch, cfg.attn_res[0]
    for i, v in enumerate(cfg.ch_mult):
        c.ch_mult[i] = v
    c.num_res_blocks, c.vis_only = cfg.num_res_blocks, int(vis_only)
    c.dtype = {"f32": 0, "bf16": 1, "f16": 2}[cfg.compute_dtype]
    return c


def densify_rna(rna, device):
    """dense tensor, or the reference's COO triple (dat, crd, ssz) (unet_ours.py:301-306)."""
    if torch.is_tensor(rna):
        return rna.to(device=device, dtype=torch.float32).contiguous()
    dat, crd, ssz = rna
    t = torch.sparse_coo_tensor(crd.long().to(device), dat.to(device=device, dtype=torch.float32), tuple(ssz))
    return t.to_dense().contiguous()


class _HipModel(torch.nn.Module):
    """Shared plumbing: owns the tm_model handle and the weight loading contract."""
    _vis_only = False

    def __init__(self, conf: PathConfig, device="cuda:0"):
        super().__init__()
        self.conf = conf
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("teramind_amd models run on a ROCm device only (no CPU fallback)")
        self._L = _lib.lib()
        self._h = C.c_void_p(0)
        cfg = _tm_config(conf, self._vis_only)
        with torch.cuda.device(self.device):
            _lib.check(self._L.tm_model_create(C.byref(cfg), C.byref(self._h)), "tm_model_create")
        self._finalized = False
        self._ws = {}                     # workspace per stream: calls on different streams never share scratch memory
        self._side = None                 # second stream of overlap_streams = 2
        # overlap_streams = 2: forward() runs the two halves of a call's images on two HIP streams, each on its half of the call's
        # workspace (images are independent; results bit-identical).  One stream's last partial round of conv workgroups and its HBM-bound block-input passes / Linears run
        # beside the other's MFMA-bound convs: one test_brn tile step 51.4 -> 50.3 ms (profiles/r03_two_streams.txt).  Off by
        # default: per-kernel durations (bench.py's roofline, rocprof summaries) are only meaningful without the overlap.
        self.overlap_streams = 1
        # lets `next(model.parameters()).device` (reference diffusion/base.py:562) work
        self._anchor = torch.nn.Parameter(torch.zeros(1, device=self.device), requires_grad=False)

    # -- weights -------------------------------------------------------------------------
    def expected_keys(self):
        n = self._L.tm_model_num_params(self._h)
        return [self._L.tm_model_param_key(self._h, i).decode() for i in range(n)]

    def load_state_dict(self, state_dict, strict: bool = True):
        """Reference contract (test_brn.py:140-147): keys as in `model.state_dict()`.
        strict=False ignores unexpected keys (as the attention driver does, test_attn.py:334);
        missing keys are always an error because the packed arena needs every tensor."""
        if self._finalized:
            raise RuntimeError("weights already loaded (the packed arena is immutable)")
        want = set(self.expected_keys())
        unexpected = [k for k in state_dict if k not in want and k != "_anchor"]
        if unexpected and strict:
            raise RuntimeError(f"Unexpected key(s) in state_dict: {unexpected[:5]}{'...' if len(unexpected) > 5 else ''}")
        for k, v in state_dict.items():
            if k not in want:
                continue
            hv = v.detach().to(device="cpu", dtype=torch.float32).contiguous()
            shp = (C.c_int64 * hv.dim())(*hv.shape)
            _lib.check(self._L.tm_model_load_param(self._h, k.encode(), C.c_void_p(hv.data_ptr()), shp, hv.dim(), 0),
                       f"tm_model_load_param({k})")
        with torch.cuda.device(self.device):
            _lib.check(self._L.tm_model_finalize(self._h), "tm_model_finalize")
        self._finalized = True
        return self

    def arena(self) -> torch.Tensor:
        """Zero-copy uint8 tensor view of the packed device weight arena (library-owned
        memory; used for the one-off RCCL broadcast of rank 0's weights)."""
        holder = _CudaArray(self._L.tm_model_arena_ptr(self._h), self._L.tm_model_arena_bytes(self._h))
        return torch.as_tensor(holder, device=self.device)

    def profile(self, on: bool):
        """Bracket every 3x3x3 conv launch (the dominant kernel) with hipEvents (measurement hook, bench.py)."""
        _lib.check(self._L.tm_profile_enable(self._h, int(on)), "tm_profile_enable")

    def profile_collect(self) -> dict:
        st = _lib.TmProfStats()
        _lib.check(self._L.tm_profile_collect(self._h, C.byref(st)), "tm_profile_collect")
        return {"launches": int(st.launches), "total_ms": st.total_ms, "nominal_flops": st.nominal_flops,
                "executed_flops": st.executed_flops, "alg_bytes": st.alg_bytes}

    def _workspace(self, nbytes: int) -> torch.Tensor:
        key = _lib.current_stream_ptr()
        key = int(getattr(key, "value", key) or 0)
        ws = self._ws.pop(key, None)
        if ws is None or ws.numel() < nbytes:
            ws = None
            while len(self._ws) >= 4:     # callers that churn through streams: keep the four most recently used workspaces
                self._ws.pop(next(iter(self._ws)))
            ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self._ws[key] = ws                # (re-)inserted last: dict order = least recently used first
        return ws

    def __del__(self):
        try:
            if self._h:
                self._L.tm_model_destroy(self._h)
                self._h = C.c_void_p(0)
        except Exception:
            pass


class _CudaArray:
    """`__cuda_array_interface__` carrier for a raw device pointer."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False),
                                         "version": 2, "strides": None}


class RnaPyramid:
    """The RNA conditioning of a forward, computed once (BeatGANsUNetModel.precompute_rna) for the many steps of a reverse
    loop over the same genes.  Opaque device buffer + the (b, p1, p2) it is valid for."""

    def __init__(self, buf: torch.Tensor, b: int, p1: int, p2: int):
        self.buf, self.b, self.p1, self.p2 = buf, b, p1, p2


class RnaLevel0:
    """Level 0 of the RNA conditioning (gene attention -> down_z -> Upsample), computed once per tile / window of a sweep
    (BeatGANsUNetModel.precompute_rna_level0) and handed to every diffusion step in place of the genes.  59 KB per patch."""

    def __init__(self, buf: torch.Tensor, b: int, p1: int, p2: int):
        self.buf, self.b, self.p1, self.p2 = buf, b, p1, p2


class BeatGANsUNetModel(_HipModel):
    def precompute_rna_level0(self, rna, b: int, imgs=None, patch_size=64) -> RnaLevel0:
        """The part of get_rna (unet_ours.py:298-310) that reads the gene counts, for `b` images whose padded patch grid is
        taken from `imgs.shape[-2:]`; pass the result as `rna=` to forward()."""
        if not self._finalized:
            raise RuntimeError("load_state_dict() must be called before precompute_rna_level0")
        H, W = imgs.shape[-2:]
        p1, p2 = H // patch_size + 1, W // patch_size + 1
        rna_d = densify_rna(rna, self.device)
        gn, zg = self.conf.gn_sz, self.conf.rna_slc * 500
        if tuple(rna_d.shape) != (b * p1 * p2, gn, gn, zg):
            raise ValueError(f"rna has shape {tuple(rna_d.shape)}, expected {(b * p1 * p2, gn, gn, zg)}")
        with torch.cuda.device(self.device):
            n = self._L.tm_rna_level0_bytes(self._h, b, p1, p2)
            buf = torch.empty(n, dtype=torch.uint8, device=self.device)
            need = self._L.tm_workspace_bytes(self._h, b, p1, p2, 0)
            ws = self._workspace(need)
            _lib.check(self._L.tm_rna_level0(self._h, _lib.ptr(rna_d), b, p1, p2, _lib.ptr(buf), n, _lib.ptr(ws), ws.numel(),
                                             _lib.current_stream_ptr()), "tm_rna_level0")
        return RnaLevel0(buf, b, p1, p2)

    def precompute_rna(self, rna, b: int, imgs=None, patch_size=64) -> RnaPyramid:
        """get_rna (unet_ours.py:298-323) for `b` images whose padded patch grid is taken from `imgs.shape[-2:]`: depends
        on the genes only, so a T-step sampler calls it once and passes the result as `rna=` to every step."""
        if not self._finalized:
            raise RuntimeError("load_state_dict() must be called before precompute_rna")
        H, W = imgs.shape[-2:]
        p1, p2 = H // patch_size + 1, W // patch_size + 1
        rna_d = densify_rna(rna, self.device)
        gn, zg = self.conf.gn_sz, self.conf.rna_slc * 500
        if tuple(rna_d.shape) != (b * p1 * p2, gn, gn, zg):
            raise ValueError(f"rna has shape {tuple(rna_d.shape)}, expected {(b * p1 * p2, gn, gn, zg)}")
        with torch.cuda.device(self.device):
            n = self._L.tm_rna_pyramid_bytes(self._h, b, p1, p2)
            buf = torch.empty(n, dtype=torch.uint8, device=self.device)
            _lib.check(self._L.tm_rna_pyramid(self._h, _lib.ptr(rna_d), b, p1, p2, _lib.ptr(buf), n, _lib.current_stream_ptr()),
                       "tm_rna_pyramid")
        return RnaPyramid(buf, b, p1, p2)

    def forward(self, x, t, rna=None, pos=None, y=None, imgs=None, cond=None, noise=None, t_cond=None,
                idx=None, index=None, do_train=False, patch_size=64, pos_random=None, random=None,
                want_pred2=False, **kwargs):
        """Same arguments as the reference forward (unet_ours.py:343-359).  `imgs` is read for
        its H, W only (:361); `pos`, `idx`, `cond`, `random` are ignored by the `ours` model.
        `want_pred2=True` additionally runs the original-patch decoder pass (training-only
        output in the reference; `pred2` is None otherwise)."""
        if do_train:
            raise NotImplementedError("training forward (do_train=True) is outside the inference hot path")
        if not self._finalized:
            raise RuntimeError("load_state_dict() must be called before forward")
        if patch_size != self.conf.patch_size:
            raise ValueError(f"patch_size {patch_size} != configured {self.conf.patch_size}")
        H, W = imgs.shape[-2:]
        p1, p2 = H // patch_size + 1, W // patch_size + 1
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        t = t.to(device=self.device, dtype=torch.int64).contiguous()
        b = t.shape[0]
        ne, nd = b * p1 * p2, b * (p1 - 1) * (p2 - 1)
        C_ = self.conf.in_channels
        if tuple(x.shape) != (ne, C_, patch_size, patch_size):
            raise ValueError(f"x has shape {tuple(x.shape)}, expected {(ne, C_, patch_size, patch_size)}")
        pyr = rna if isinstance(rna, (RnaPyramid, RnaLevel0)) else None
        if pyr is not None:
            if (pyr.b, pyr.p1, pyr.p2) != (b, p1, p2):
                raise ValueError(f"{type(pyr).__name__} was computed for (b, p1, p2) = {(pyr.b, pyr.p1, pyr.p2)}, this call has {(b, p1, p2)}")
        else:
            rna_d = densify_rna(rna, self.device)
            gn, zg = self.conf.gn_sz, self.conf.rna_slc * 500
            if tuple(rna_d.shape) != (ne, gn, gn, zg):
                raise ValueError(f"rna has shape {tuple(rna_d.shape)}, expected {(ne, gn, gn, zg)}")
        pred = torch.empty((nd, C_, patch_size, patch_size), dtype=torch.float32, device=self.device)
        pred2 = torch.empty_like(x) if want_pred2 else None

        def run(i0, i1, ws=None):         # images i0 .. i1 - 1 of the call, on the current stream, into their slices of pred / pred2
            bb, pe, pd = i1 - i0, p1 * p2, (p1 - 1) * (p2 - 1)
            xs, ts, ps_ = x[i0 * pe:i1 * pe], t[i0:i1], pred[i0 * pd:i1 * pd]
            p2s = pred2[i0 * pe:i1 * pe] if want_pred2 else None
            if ws is None:
                ws = self._workspace(self._L.tm_workspace_bytes(self._h, bb, p1, p2, int(want_pred2)))
            if isinstance(pyr, RnaLevel0):                                    # patch-major: an image's slice is contiguous
                per = pyr.buf.numel() // (pyr.b * pe)
                l0 = pyr.buf[i0 * pe * per:i1 * pe * per]
                _lib.check(self._L.tm_unet_forward_level0(self._h, _lib.ptr(xs), _lib.ptr(ts), _lib.ptr(l0), l0.numel(), bb, p1,
                                                          p2, _lib.ptr(ps_), _lib.ptr(p2s), _lib.ptr(ws), ws.numel(),
                                                          _lib.current_stream_ptr()), "tm_unet_forward_level0")
            elif pyr is not None:
                _lib.check(self._L.tm_unet_forward_rna(self._h, _lib.ptr(xs), _lib.ptr(ts), _lib.ptr(pyr.buf), pyr.buf.numel(), bb, p1, p2,
                                                       _lib.ptr(ps_), _lib.ptr(p2s), _lib.ptr(ws), ws.numel(),
                                                       _lib.current_stream_ptr()), "tm_unet_forward_rna")
            else:
                rs = rna_d[i0 * pe:i1 * pe]
                _lib.check(self._L.tm_unet_forward(self._h, _lib.ptr(xs), _lib.ptr(ts), _lib.ptr(rs), bb, p1, p2,
                                                   _lib.ptr(ps_), _lib.ptr(p2s), _lib.ptr(ws), ws.numel(),
                                                   _lib.current_stream_ptr()), "tm_unet_forward")

        with torch.cuda.device(self.device):
            # the RnaPyramid buffer is opaque (several tensors per call): only level-0 / dense-gene calls are split
            if self.overlap_streams >= 2 and b >= 2 and not isinstance(pyr, RnaPyramid):
                if self._side is None:
                    self._side = torch.cuda.Stream(device=self.device)
                cur, half = torch.cuda.current_stream(self.device), (b + 1) // 2
                # ONE allocation (the caller's stream's workspace) cut in two: the side stream touches its part only between the
                # two wait_stream calls, and the memory of a call does not grow with the number of streams
                n0 = self._L.tm_workspace_bytes(self._h, half, p1, p2, int(want_pred2))
                n1 = self._L.tm_workspace_bytes(self._h, b - half, p1, p2, int(want_pred2))
                n0 = (n0 + 4095) // 4096 * 4096
                ws = self._workspace(n0 + n1)
                self._side.wait_stream(cur)                                   # inputs were produced on the caller's stream
                run(0, half, ws[:n0])
                with torch.cuda.stream(self._side):
                    run(half, b, ws[n0:n0 + n1])
                cur.wait_stream(self._side)                                   # the caller's stream sees both halves complete
            else:
                run(0, b)
        return AutoencReturn(pred=pred, pred2=pred2, cond=cond)


class GeneAttnModel(_HipModel):
    """Attention-map model: forward(x, t, rna, imgs) -> (attn[4,B,G,G], rna_h[:, :, 1:-1])."""
    _vis_only = True

    def forward(self, x=None, t=None, rna=None, imgs=None, **kwargs):
        if not self._finalized:
            raise RuntimeError("load_state_dict() must be called before forward")
        rna_d = densify_rna(rna, self.device)
        B, gn = rna_d.shape[0], self.conf.gn_sz
        G, zs = self.conf.rna_num, self.conf.rna_slc
        if tuple(rna_d.shape) != (B, gn, gn, zs * 500):
            raise ValueError(f"rna has shape {tuple(rna_d.shape)}, expected {(B, gn, gn, zs * 500)}")
        attn = torch.empty((4, B, G, G), dtype=torch.float32, device=self.device)
        mid = torch.empty((B, G, zs - 2, gn, gn), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            ws = self._workspace(self._L.tm_gene_attn_workspace_bytes(self._h, B))
            _lib.check(self._L.tm_gene_attn(self._h, _lib.ptr(rna_d), B, _lib.ptr(attn), _lib.ptr(mid),
                                            _lib.ptr(ws), ws.numel(), _lib.current_stream_ptr()), "tm_gene_attn")
        return attn, mid


def make_model(conf: PathConfig, device="cuda:0", state_dict=None, vis_only=False):
    m = (GeneAttnModel if vis_only else BeatGANsUNetModel)(conf, device)
    if state_dict is not None:
        m.load_state_dict(state_dict, strict=not vis_only)
    return m


__all__ = ["AutoencReturn", "RnaPyramid", "RnaLevel0", "BeatGANsUNetModel", "GeneAttnModel", "make_model", "param_spec"]
