"""Training slice, whole model: the training-mode forward of the patch UNet (do_train: p1 = p2 = 2,
both decoder passes, model/unet_ours.py:343-426) and its BACKWARD, composed from the HIP kernels of csrc/tm_train.hip and the
forward conv / prep kernels -- no torch.autograd, no torch arithmetic on the model path.

The reference trains through torch.autograd over stock modules (experiment.py:121-193 -> diffusion/base.py:181-289 ->
model/unet_ours.py).  Here `UNetTrain.forward` records a tape of ops, each with a hand-written adjoint:

    conv3 / conv1     tm_op_conv_mfma          | tm_op_conv_dgrad (same kernel, transposed + flipped weights), tm_op_conv_wgrad
                      ((1,3,3) and the down_z (kz,3,3) kernels run embedded in the 3x3x3 'same' conv, sliced afterwards)
    SiLU(RMSNorm * w [* (1 + scale) + shift])   tm_op_prep_train | tm_op_prep_bwd     (ResBlock in / out layers, the head)
    nearest x2 / AvgPool(1,2,2)                 tm_op_resample   | the other mode, x 4 or / 4 (tm_op_ew 7 / 8)
    AttnBlock with gene cross-attention         training.AttnBlockTrain (modulate(norm), windowed attention core, MLP, gates)
    Linears over rows (time embedding, emb_layers, the gene-gene AttnBlock)   tm_op_gemm_f32 in its three roles
    row RMSNorm / softmax of the gene-gene block                              tm_op_rows
    SiLU / GELU / adds                                                        tm_op_ew

Channel concatenation, the half-patch collage (model/unet_ours.py:325-341) and the z slice of down_z are re-indexings of
device tensors (torch views / cat / pad: no arithmetic); the loss and d(loss)/d(pred) are a handful of elementwise torch
ops on the two [n, C, ps, ps] predictions (diffusion/base.py:272-288).  The model runs as in `.eval()` with gradients --
ResBlock dropout (p = 0.1, config_parm.py:46) is the identity here; `training.ResBlockTrain` covers the dropout forward /
backward with a supplied mask.  Functional, not tuned: every op synchronises, weights are re-packed per call.
"""
import ctypes as C
import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from . import _lib
from .config import PathConfig
from .training import AttnBlockTrain, _cb8, _host, _hp, _ncdhw


class _V:
    """A tape value: `t` = device tensor (CB8 [N, ceil(C/8), Z, S, S, 8] when `C` is set, plain otherwise), `g` = dL/dt."""
    __slots__ = ("t", "C", "g")

    def __init__(self, t, C=None):
        self.t, self.C, self.g = t, C, None


def block_plan(cfg: PathConfig):
    """The block list the constructor implies (model/unet_ours.py:134-269): encoder entries (level, ops, concat-rna-first),
    the middle block, decoder entries (level, ops); an op is (kind, prefix, mode)."""
    L = len(cfg.ch_mult)
    res, k, enc = cfg.patch_size, 1, []
    for lvl in range(L):
        for _ in range(cfg.num_res_blocks):
            ops = [("res", f"input_blocks.{k}.0", "same")]
            if res in cfg.attn_res:
                ops.append(("attn", f"input_blocks.{k}.1", ""))
            enc.append((lvl, ops, True))
            k += 1
        if lvl != L - 1:
            res //= 2
            enc.append((lvl + 1, [("res", f"input_blocks.{k}.0", "down")], False))
            k += 1
    mid = [("res", "middle_block.0", "same"), ("attn", "middle_block.1", ""), ("res", "middle_block.2", "same")]
    dec, k = [], 0
    for lvl in reversed(range(L)):
        for i in range(cfg.num_res_blocks + 1):
            ops, nxt = [("res", f"output_blocks.{k}.0", "same")], 1
            if res in cfg.attn_res:
                ops.append(("attn", f"output_blocks.{k}.1", ""))
                nxt = 2
            if lvl and i == cfg.num_res_blocks:
                res *= 2
                ops.append(("res", f"output_blocks.{k}.{nxt}", "up"))
            dec.append((lvl, ops))
            k += 1
    return enc, mid, dec


class UNetTrain:
    """forward(x_p, t_map, rna_dense, b) -> (pred, pred2); backward(dpred, dpred2) -> {state_dict key: gradient (host fp32)}.
    `state` = the reference state_dict (host tensors / arrays)."""

    def __init__(self, cfg: PathConfig, state: Dict[str, "object"], device="cuda:0"):
        self.cfg = cfg
        self.dev = torch.device(device)
        self.W = {k: _host(torch.as_tensor(v)) for k, v in state.items()}
        self._Wd: Dict[str, torch.Tensor] = {}
        self.tape: List = []
        self.grads: Dict[str, torch.Tensor] = {}
        self._one = None
        if cfg.down_z_kernel not in (1, 3):
            raise NotImplementedError("UNetTrain: down_z kernels of depth 1 and 3 (rna_slc 1, 4)")

    # ------------------------------------------------------------------------------------------------------------------
    # plumbing
    # ------------------------------------------------------------------------------------------------------------------
    def _st(self):
        return _lib.current_stream_ptr()

    def _wd(self, key):
        if key not in self._Wd:
            self._Wd[key] = self.W[key].to(self.dev).contiguous()
        return self._Wd[key]

    def _gacc(self, key, g):
        g = g.reshape(self.W[key].shape)
        self.grads[key] = g.clone() if key not in self.grads else self.grads[key] + g

    def _acc(self, v: _V, g: torch.Tensor):
        v.g = g if v.g is None else self._ew(6, v.g, g)

    def _ew(self, op, a, b=None, c=None, two=False):
        o1 = torch.empty_like(a)
        o2 = torch.empty_like(a) if two else None
        _lib.check(_lib.lib().tm_op_ew(op, _lib.ptr(a), _lib.ptr(b), _lib.ptr(c), _lib.ptr(o1), _lib.ptr(o2), a.numel(), self._st()), "tm_op_ew")
        return (o1, o2) if two else o1

    @staticmethod
    def _geo(cb):
        return cb.shape[0], cb.shape[2], cb.shape[3]                    # N, Z, S

    # ------------------------------------------------------------------------------------------------------------------
    # tape ops on CB8 values
    # ------------------------------------------------------------------------------------------------------------------
    def conv(self, x: _V, key: str) -> _V:
        """Conv3d with the reference weight `key`.weight [Co, Ci, kz, ky, kx]: 3x3x3 pad 1, 1x1x1, or (1,3,3) pad (0,1,1)
        embedded into the middle z slice of a 3x3x3 kernel."""
        w = self.W[key + ".weight"]
        b = self.W[key + ".bias"]
        co, ci = w.shape[:2]
        assert ci == x.C, (key, ci, x.C)
        ks = 1 if w.shape[2:] == (1, 1, 1) else 3
        embed = tuple(w.shape[2:]) == (1, 3, 3)
        if embed:
            wf = torch.zeros((co, ci, 3, 3, 3), dtype=torch.float32)
            wf[:, :, 1] = w[:, :, 0]
        else:
            wf = w.contiguous()
        N, Z, S = self._geo(x.t)
        y = torch.zeros((N, (co + 7) // 8, Z, S, S, 8), dtype=torch.float32, device=self.dev)
        _lib.check(_lib.lib().tm_op_conv_mfma(_lib.ptr(x.t), _hp(wf), _hp(b), _lib.ptr(y), N, ci, co, Z, S, ks, 0, 0, 0, self._st()), "tm_op_conv_mfma")
        out = _V(y, co)

        def bwd():
            g = out.g
            dw = torch.empty((co, ci) + ((3, 3, 3) if ks == 3 else (1, 1, 1)), dtype=torch.float32)
            db = torch.empty((co,), dtype=torch.float32)
            _lib.check(_lib.lib().tm_op_conv_wgrad(_lib.ptr(x.t), _lib.ptr(g), _hp(dw), _hp(db), N, ci, co, Z, S, ks, self._st()), "tm_op_conv_wgrad")
            self._gacc(key + ".weight", dw[:, :, 1:2].contiguous() if embed else dw)
            self._gacc(key + ".bias", db)
            dx = torch.zeros_like(x.t)
            _lib.check(_lib.lib().tm_op_conv_dgrad(_lib.ptr(g), _hp(wf), _lib.ptr(dx), N, ci, co, Z, S, ks, self._st()), "tm_op_conv_dgrad")
            self._acc(x, dx)
        self.tape.append(bwd)
        return out

    def prep(self, x: _V, key: str, ss: Optional[_V] = None, per_image: int = 1) -> _V:
        """SiLU(RMSNorm_C(x) * w [* (1 + scale[img]) + shift[img]]); ss = plain [nimg, 2 C] value (scale | shift)."""
        nw = self.W[key].reshape(-1)
        N, Z, S = self._geo(x.t)
        sc = sh = None
        if ss is not None:
            h = ss.t.to("cpu")
            sc, sh = h[:, :x.C].contiguous(), h[:, x.C:].contiguous()
        y = torch.empty_like(x.t)
        _lib.check(_lib.lib().tm_op_prep_train(_lib.ptr(x.t), _hp(nw), _hp(sc), _hp(sh), None, 1.0, per_image, _lib.ptr(y), N, x.C, Z, S,
                                               self._st()), "tm_op_prep_train")
        out = _V(y, x.C)

        def bwd():
            dx = torch.empty_like(x.t)
            nimg = (N + per_image - 1) // per_image
            dw = torch.empty((x.C,), dtype=torch.float32)
            dsc = torch.empty((nimg, x.C), dtype=torch.float32) if ss is not None else None
            dsh = torch.empty((nimg, x.C), dtype=torch.float32) if ss is not None else None
            _lib.check(_lib.lib().tm_op_prep_bwd(_lib.ptr(x.t), _lib.ptr(out.g), _hp(nw), _hp(sc), _hp(sh), None, 1.0, per_image, _lib.ptr(dx),
                                                 _hp(dw), _hp(dsc), _hp(dsh), N, x.C, Z, S, self._st()), "tm_op_prep_bwd")
            self._gacc(key, dw)
            self._acc(x, dx)
            if ss is not None:
                self._acc(ss, torch.cat([dsc, dsh], dim=1).to(self.dev))
        self.tape.append(bwd)
        return out

    def _resample_raw(self, t, C_, mode):
        N, Z, S = self._geo(t)
        So = S * 2 if mode == 1 else S // 2
        y = torch.empty((N, t.shape[1], Z, So, So, 8), dtype=torch.float32, device=self.dev)
        _lib.check(_lib.lib().tm_op_resample(_lib.ptr(t), _lib.ptr(y), N, C_, Z, So, mode, self._st()), "tm_op_resample")
        return y

    def resample(self, x: _V, mode: int) -> _V:
        """mode 1: nearest x2 on (h, w); mode 2: AvgPool(1,2,2)."""
        out = _V(self._resample_raw(x.t, x.C, mode), x.C)

        def bwd():
            # up2^T = 4 * avgpool;  avgpool^T = up2 / 4
            self._acc(x, self._ew(7 if mode == 1 else 8, self._resample_raw(out.g, x.C, 2 if mode == 1 else 1)))
        self.tape.append(bwd)
        return out

    def silu(self, x: _V) -> _V:
        out = _V(self._ew(4, x.t), x.C)
        self.tape.append(lambda: self._acc(x, self._ew(5, out.g, x.t)))
        return out

    def add(self, a: _V, b: _V) -> _V:
        out = _V(self._ew(6, a.t, b.t), a.C)

        def bwd():
            self._acc(a, out.g)
            self._acc(b, out.g)
        self.tape.append(bwd)
        return out

    def cat(self, vs: List[_V]) -> _V:
        assert all(v.C % 8 == 0 for v in vs[:-1]), "only the last operand of a channel concat may have a partial block"
        out = _V(torch.cat([v.t for v in vs], dim=1).contiguous(), sum(v.C for v in vs))

        def bwd():
            o = 0
            for v in vs:
                nb = v.t.shape[1]
                self._acc(v, out.g[:, o:o + nb].contiguous())
                o += nb
        self.tape.append(bwd)
        return out

    def collage(self, x: _V, b: int, p1: int, p2: int) -> _V:
        """(b p1 p2) patches -> (b (p1-1) (p2-1)) half-patch-shifted patches (model/unet_ours.py:325-341): a re-indexing."""
        n, Cb, Z, S, _, _ = x.t.shape
        hp, q1, q2 = S // 2, p1 - 1, p2 - 1
        img = x.t.reshape(b, p1, p2, Cb, Z, S, S, 8).permute(0, 3, 4, 1, 5, 2, 6, 7).reshape(b, Cb, Z, p1 * S, p2 * S, 8)
        img = img[:, :, :, hp:p1 * S - hp, hp:p2 * S - hp]
        out = _V(img.reshape(b, Cb, Z, q1, S, q2, S, 8).permute(0, 3, 5, 1, 2, 4, 6, 7).reshape(b * q1 * q2, Cb, Z, S, S, 8).contiguous(), x.C)

        def bwd():
            g = out.g.reshape(b, q1, q2, Cb, Z, S, S, 8).permute(0, 3, 4, 1, 5, 2, 6, 7).reshape(b, Cb, Z, q1 * S, q2 * S, 8)
            g = F.pad(g, (0, 0, hp, hp, hp, hp))
            self._acc(x, g.reshape(b, Cb, Z, p1, S, p2, S, 8).permute(0, 3, 5, 1, 2, 4, 6, 7).reshape(n, Cb, Z, S, S, 8).contiguous())
        self.tape.append(bwd)
        return out

    def attn_block(self, x: _V, cond: _V, pfx: str) -> _V:
        blk = AttnBlockTrain({k[len(pfx) + 1:]: v for k, v in self.W.items() if k.startswith(pfx + ".")}, self.dev)
        out = _V(blk.forward_cb(x.t, cond.t), x.C)

        def bwd():
            dx, dcond, g = blk.backward_cb(out.g)
            self._acc(x, dx)
            self._acc(cond, dcond)
            for k, v in g.items():
                self._gacc(f"{pfx}.{k}", v)
        self.tape.append(bwd)
        return out

    def res_block(self, x: _V, ste: _V, pfx: str, mode: str, per_image: int) -> _V:
        """ResBlock._forward + apply_conditions (model/MBAblocks.py:237-299, 302-368); ste = SiLU(time embedding) [b, E]."""
        a = self.prep(x, f"{pfx}.in_layers.0.weight")
        xs = x
        if mode == "up":
            a, xs = self.resample(a, 1), self.resample(x, 1)
        elif mode == "down":
            a, xs = self.resample(a, 2), self.resample(x, 2)
        h1 = self.conv(a, f"{pfx}.in_layers.2")
        ss = self.linear(ste, f"{pfx}.emb_layers.1")
        d = self.prep(h1, f"{pfx}.out_layers.0.weight", ss, per_image)
        h2 = self.conv(d, f"{pfx}.out_layers.3")
        if f"{pfx}.skip_connection.weight" in self.W:
            xs = self.conv(xs, f"{pfx}.skip_connection")
        return self.add(xs, h2)

    # ------------------------------------------------------------------------------------------------------------------
    # tape ops on plain [rows, D] values
    # ------------------------------------------------------------------------------------------------------------------
    def _gemm(self, A, B, Cm, M, N, K, st, batch=1, bias=None, bias_mode=0, alpha=1.0, accumulate=0):
        arr = (C.c_long * 9)(*st)
        _lib.check(_lib.lib().tm_op_gemm_f32(_lib.ptr(A), _lib.ptr(B), _lib.ptr(bias), _lib.ptr(Cm), M, N, K, C.cast(arr, C.c_void_p), batch,
                                             bias_mode, accumulate, alpha, self._st()), "tm_op_gemm_f32")

    def linear(self, x: _V, key: str) -> _V:
        """y = x W^T + b on [rows, Din] (nn.Linear)."""
        w, bias = self._wd(key + ".weight"), self._wd(key + ".bias")
        dout, din = w.shape
        rows = x.t.numel() // din
        y = torch.empty((rows, dout), dtype=torch.float32, device=self.dev)
        self._gemm(x.t, w, y, rows, dout, din, (din, 1, 1, din, dout, 1, 0, 0, 0), bias=bias, bias_mode=1)
        out = _V(y)

        def bwd():
            g = out.g.contiguous()
            dx = torch.empty((rows, din), dtype=torch.float32, device=self.dev)
            self._gemm(g, w, dx, rows, din, dout, (dout, 1, din, 1, din, 1, 0, 0, 0))
            dw = torch.empty((dout, din), dtype=torch.float32, device=self.dev)
            self._gemm(g, x.t, dw, dout, din, rows, (1, dout, din, 1, din, 1, 0, 0, 0))
            if self._one is None:
                self._one = torch.ones(1, dtype=torch.float32, device=self.dev)
            db = torch.empty((dout,), dtype=torch.float32, device=self.dev)
            self._gemm(g, self._one, db, dout, 1, rows, (1, dout, 0, 0, 1, 0, 0, 0, 0))
            self._gacc(key + ".weight", dw.cpu())
            self._gacc(key + ".bias", db.cpu())
            self._acc(x, dx.reshape(x.t.shape))
        self.tape.append(bwd)
        return out

    def rms_rows(self, x: _V, key: str) -> _V:
        w = self._wd(key)
        D = w.numel()
        rows = x.t.numel() // D
        y = torch.empty_like(x.t)
        _lib.check(_lib.lib().tm_op_rows(0, _lib.ptr(x.t), _lib.ptr(w), None, _lib.ptr(y), None, rows, D, self._st()), "tm_op_rows")
        out = _V(y)

        def bwd():
            dx = torch.empty_like(x.t)
            dw = torch.empty((D,), dtype=torch.float32, device=self.dev)
            _lib.check(_lib.lib().tm_op_rows(1, _lib.ptr(x.t), _lib.ptr(w), _lib.ptr(out.g.contiguous()), _lib.ptr(dx), _lib.ptr(dw), rows, D,
                                             self._st()), "tm_op_rows")
            self._gacc(key, dw.cpu())
            self._acc(x, dx)
        self.tape.append(bwd)
        return out

    def act_rows(self, x: _V, fwd_op: int) -> _V:
        """fwd_op 2 = GELU(tanh), 4 = SiLU on a plain tensor."""
        out = _V(self._ew(fwd_op, x.t))
        self.tape.append(lambda: self._acc(x, self._ew(fwd_op + 1, out.g.contiguous(), x.t)))
        return out

    def gene_attention(self, tok: torch.Tensor, n: int, G: int, D: int) -> _V:
        """The gene-gene AttnBlock body (gene_trans=False, model/MBAblocks.py:492-501, 551-601: k = q, q_norm on both, no
        residuals): tok [n * G, D] (no gradient) -> mlp output [n * G, D]."""
        p = "rna_blocks.0.0"
        x = _V(tok)
        q = self.linear(x, f"{p}.attn.q")
        v = self.linear(x, f"{p}.attn.v")
        qn = self.rms_rows(q, f"{p}.attn.q_norm.weight")
        logits = torch.empty((n, G, G), dtype=torch.float32, device=self.dev)
        self._gemm(qn.t, qn.t, logits, G, G, D, (D, 1, 1, D, G, 1, G * D, G * D, G * G), batch=n, alpha=1.0 / D)
        prob = torch.empty_like(logits)
        _lib.check(_lib.lib().tm_op_rows(2, _lib.ptr(logits), None, None, _lib.ptr(prob), None, n * G, G, self._st()), "tm_op_rows")
        o1t = torch.empty((n * G, D), dtype=torch.float32, device=self.dev)
        self._gemm(prob, v.t, o1t, G, D, G, (G, 1, D, 1, D, 1, G * G, G * D, G * D), batch=n)
        o1 = _V(o1t)

        def bwd():
            g = o1.g.contiguous()
            # o1 = P v:  dv = P^T g,  dP = g v^T
            dv = torch.empty_like(v.t)
            self._gemm(prob, g, dv, G, D, G, (1, G, D, 1, D, 1, G * G, G * D, G * D), batch=n)
            self._acc(v, dv)
            dP = torch.empty_like(prob)
            self._gemm(g, v.t, dP, G, G, D, (D, 1, 1, D, G, 1, G * D, G * D, G * G), batch=n)
            dS = torch.empty_like(prob)
            _lib.check(_lib.lib().tm_op_rows(3, _lib.ptr(prob), None, _lib.ptr(dP), _lib.ptr(dS), None, n * G, G, self._st()), "tm_op_rows")
            # logits = qn qn^T / D:  dqn = (dS + dS^T) qn / D
            dqn = torch.empty_like(qn.t)
            self._gemm(dS, qn.t, dqn, G, D, G, (G, 1, D, 1, D, 1, G * G, G * D, G * D), batch=n, alpha=1.0 / D)
            self._gemm(dS, qn.t, dqn, G, D, G, (1, G, D, 1, D, 1, G * G, G * D, G * D), batch=n, alpha=1.0 / D, accumulate=1)
            self._acc(qn, dqn)
        self.tape.append(bwd)
        o2 = self.linear(o1, f"{p}.attn.proj")
        o3 = self.rms_rows(o2, f"{p}.norm2.weight")
        o4 = self.linear(o3, f"{p}.mlp.fc1")
        o5 = self.act_rows(o4, 2)
        return self.linear(o5, f"{p}.mlp.fc2")

    def rows_to_cb8(self, x: _V, shape, C_) -> _V:
        """plain [n, C, Z, h, w] (any view of it) -> CB8."""
        out = _V(_cb8(x.t.reshape(shape)), C_)
        self.tape.append(lambda: self._acc(x, _ncdhw(out.g, C_).reshape(x.t.shape)))
        return out

    def zslice(self, x: _V, z0: int, z1: int) -> _V:
        out = _V(x.t[:, :, z0:z1].contiguous(), x.C)

        def bwd():
            g = torch.zeros_like(x.t)
            g[:, :, z0:z1] = out.g
            self._acc(x, g)
        self.tape.append(bwd)
        return out

    # ------------------------------------------------------------------------------------------------------------------
    # the model
    # ------------------------------------------------------------------------------------------------------------------
    def rna_pyramid(self, rna: torch.Tensor) -> List[_V]:
        """get_rna (model/unet_ours.py:277-323): dense [n, gh, gw, zs * 500] -> the four gene-condition levels."""
        cfg = self.cfg
        n, gh, gw, zg = rna.shape
        zs, G = zg // 500, cfg.rna_num
        if G > 500:
            raise ValueError("rna_num > 500")
        rna_h = rna.to(self.dev).float().reshape(n, gh, gw, zs, 500).permute(0, 4, 3, 1, 2)[:, :G].contiguous()
        D = zs * gh * gw
        tok = self.gene_attention(rna_h.reshape(n * G, D), n, G, D)
        x = self.rows_to_cb8(tok, (n, G, zs, gh, gw), G)
        x = self.conv(x, "rna_blocks.0.0.down_z")
        if cfg.down_z_kernel == 3:
            x = self.zslice(x, 1, zs - 1)                                  # padding 0 along z: the interior planes of the 'same' conv
        out = [self.resample(x, 1)]
        for rid in (1, 2, 3):
            x = self.conv(self.silu(out[-1]), f"rna_blocks.{rid}.1")
            out.append(self.resample(x, 1))
        return out

    def forward(self, x_p: torch.Tensor, t_map: torch.Tensor, rna: torch.Tensor, b: int):
        """x_p [b * 4, n_stain * z, ps, ps] (the four patches of each image's 2 x 2 window), t_map [b] model-scale timesteps,
        rna dense [b * 4, gn, gn, zs * 500]  ->  pred [b, C, ps, ps] (the centre collage patch), pred2 [b * 4, C, ps, ps]."""
        cfg = self.cfg
        self.tape, self.grads = [], {}
        p1 = p2 = 2
        ne, nd = p1 * p2, (p1 - 1) * (p2 - 1)
        z, L = cfg.z_size, len(cfg.ch_mult)
        assert x_p.shape[0] == b * ne
        # time embedding (model/unet_ours.py:368-374, 442-476): sinusoid -> Linear -> SiLU -> Linear; the blocks take SiLU of it
        half = cfg.net_ch // 2
        freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
        args = t_map.cpu()[:, None].float() * freqs[None]
        sin = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
        if cfg.net_ch % 2:
            sin = torch.cat([sin, torch.zeros_like(sin[:, :1])], dim=-1)
        te = self.linear(self.act_rows(self.linear(_V(sin.to(self.dev).contiguous()), "time_embed.time_embed.0"), 4), "time_embed.time_embed.2")
        ste = self.act_rows(te, 4)
        rna_l = self.rna_pyramid(rna)
        enc, mid, dec = block_plan(cfg)

        def run(ops, h, cond, per_image):
            for kind, pfx, mode in ops:
                h = self.res_block(h, ste, pfx, mode, per_image) if kind == "res" else self.attn_block(h, cond, pfx)
            return h

        x5 = x_p.to(self.dev).float().reshape(x_p.shape[0], cfg.n_stain, z, x_p.shape[-2], x_p.shape[-1])
        h = self.conv(_V(_cb8(x5), cfg.n_stain), "input_blocks.0.0")
        skips = [[] for _ in range(L)]
        skips[0].append(h)
        for lvl, ops, cat_rna in enc:
            cond = rna_l[L - 1 - lvl]
            if cat_rna:
                h = self.cat([h, cond])
            h = run(ops, h, cond, ne)
            skips[lvl].append(h)
        h = run(mid, self.cat([h, rna_l[0]]), rna_l[0], ne)

        def decode(use_collage: bool):
            tf = (lambda a: self.collage(a, b, p1, p2)) if use_collage else (lambda a: a)
            per_image = nd if use_collage else ne
            hd = tf(h)
            stacks = [list(s) for s in skips]
            for lvl, ops in dec:
                cond = tf(rna_l[L - 1 - lvl])
                hd = run(ops, self.cat([hd, tf(stacks[lvl].pop()), cond]), cond, per_image)
            o = self.conv(self.prep(hd, "out.0.weight"), "out.2")
            return o

        self._pred, self._pred2 = decode(True), decode(False)
        S = cfg.patch_size
        to_img = lambda v: _ncdhw(v.t, cfg.n_stain).reshape(v.t.shape[0], cfg.n_stain * z, S, S)
        return to_img(self._pred), to_img(self._pred2)

    def backward(self, dpred: torch.Tensor, dpred2: torch.Tensor) -> Dict[str, torch.Tensor]:
        cfg = self.cfg
        z, S = cfg.z_size, cfg.patch_size
        for v, g in ((self._pred, dpred), (self._pred2, dpred2)):
            v.g = _cb8(g.to(self.dev).float().reshape(g.shape[0], cfg.n_stain, z, S, S))
        for bwd in reversed(self.tape):
            bwd()
        self.tape = []
        return self.grads


def training_loss_and_grads(net: UNetTrain, sampler, x_start, r_start, t, loss_mask, noise, crop_index: Tuple[int, int], patch_size: int = 64,
                            loss_type: str = "mse"):
    """One training objective evaluation with gradients: GaussianDiffusionBeatGans.training_losses (diffusion/base.py:181-289,
    restated forward-only in diffusion.SpacedDiffusionBeatGans.training_losses) + the backward of the whole model.
    Returns (loss, grads)."""
    from .diffusion import sparse_repatch
    from .unet import densify_rna
    dev = net.dev
    halfp = patch_size // 2
    x_start, noise, loss_mask = x_start.to(dev), noise.to(dev), loss_mask.to(dev)
    t = t.to(dev).long()
    x_t = sampler.q_sample(x_start, t.repeat_interleave(x_start.shape[0] // t.shape[0]), noise) * loss_mask
    ix, iy = crop_index
    dat, crd, ssz = r_start
    r_size = patch_size // (x_start.shape[2] // ssz[1])
    crd = crd.long()
    keep = (ix * r_size <= crd[1]) & (crd[1] < (ix + 2) * r_size) & (iy * r_size <= crd[2]) & (crd[2] < (iy + 2) * r_size)
    dat, crd = dat[keep], crd[:, keep].clone()
    crd[1] -= ix * r_size
    crd[2] -= iy * r_size
    dat2, crd2, ssz2 = sparse_repatch((dat, crd, (ssz[0], 2 * r_size, 2 * r_size, ssz[-1])), r_size)
    rna = densify_rna((dat2, crd2, ssz2), dev)
    sl = (slice(None), slice(None), slice(ix * patch_size, (ix + 2) * patch_size), slice(iy * patch_size, (iy + 2) * patch_size))

    def tiles2(a):
        b_, c_ = a.shape[:2]
        return a.reshape(b_, c_, 2, patch_size, 2, patch_size).permute(0, 2, 4, 1, 3, 5).reshape(b_ * 4, c_, patch_size, patch_size)

    x_p, n_p, m_p = tiles2(x_t[sl]), tiles2(noise[sl]), tiles2(loss_mask[sl])
    b_ = t.shape[0]
    tm = torch.tensor(sampler.timestep_map, dtype=torch.int64, device=dev)[t]
    pred, pred2 = net.forward(x_p, tm, rna, b_)
    n_img = n_p.reshape(b_, 2, 2, -1, patch_size, patch_size).permute(0, 3, 1, 4, 2, 5).reshape(b_, -1, 2 * patch_size, 2 * patch_size)
    noise_shift = n_img[:, :, halfp:-halfp, halfp:-halfp]
    d1, d2 = noise_shift - pred, n_p - pred2
    per1, per2 = d1[0].numel() * d1.shape[0], d2[0].numel() * d2.shape[0]
    if loss_type == "mse":
        loss = (d1 ** 2).reshape(b_, -1).mean(1).mean() + ((d2 ** 2) * m_p).reshape(d2.shape[0], -1).mean(1).mean()
        g1, g2 = -2.0 * d1 / per1, -2.0 * d2 * m_p / per2
    else:
        loss = d1.abs().reshape(b_, -1).mean(1).mean() + (d2.abs() * m_p).reshape(d2.shape[0], -1).mean(1).mean()
        g1, g2 = -torch.sign(d1) / per1, -torch.sign(d2) * m_p / per2
    grads = net.backward(g1.contiguous(), g2.contiguous())
    return float(loss), grads


class AdamTrainer:
    """The optimizer half of the reference's training step (experiment.py:207-219, 394-414): clip_grad_norm_(max_norm =
    conf.grad_clip = 1) over all parameters, then torch.optim.Adam(lr = 2e-5, weight_decay = 0; config_parm.py:48, config.py:
    76-88) -- on one flat fp32 device arena (parameters, gradients, both moments), two kernels per step (tm_op_sumsq,
    tm_op_adam).  `accum` micro-batches' gradients are averaged before the step (conf.accum_batches, config_parm.py:45)."""

    def __init__(self, net: UNetTrain, lr: float = 2e-5, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 grad_clip: float = 1.0):
        self.net, self.lr, self.betas, self.eps, self.wd, self.clip = net, lr, betas, eps, weight_decay, grad_clip
        self.keys = list(net.W)
        self.off, n = {}, 0
        for k in self.keys:
            self.off[k] = n
            n += net.W[k].numel()
        self.n = n
        dev = net.dev
        self.p = torch.cat([net.W[k].reshape(-1) for k in self.keys]).to(dev)
        self.g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.m, self.v = torch.zeros_like(self.g), torch.zeros_like(self.g)
        self.t, self._micro = 0, 0

    def accumulate(self, grads: Dict[str, torch.Tensor]):
        """Adds one micro-batch's gradients (host tensors keyed like the state_dict) into the arena."""
        flat = torch.cat([grads[k].reshape(-1) for k in self.keys]).to(self.net.dev)
        self.g = flat if self._micro == 0 else self.net._ew(6, self.g, flat)
        self._micro += 1

    def step(self) -> Dict[str, float]:
        """clip + Adam on the accumulated gradients; writes the new parameters back into the model.  Returns the total
        gradient norm (of the averaged gradient) and the clip coefficient, as clip_grad_norm_ computes them."""
        if self._micro == 0:
            raise RuntimeError("AdamTrainer.step: no gradients accumulated")
        ss = C.c_float(0.0)
        _lib.check(_lib.lib().tm_op_sumsq(_lib.ptr(self.g), self.n, C.cast(C.byref(ss), C.c_void_p), self.net._st()), "tm_op_sumsq")
        avg = 1.0 / self._micro
        total = math.sqrt(ss.value) * avg
        coef = min(1.0, self.clip / (total + 1e-6)) if self.clip > 0 else 1.0
        self.t += 1
        _lib.check(_lib.lib().tm_op_adam(_lib.ptr(self.p), _lib.ptr(self.g), _lib.ptr(self.m), _lib.ptr(self.v), self.n, self.lr, self.betas[0],
                                         self.betas[1], self.eps, self.wd, self.t, coef * avg, self.net._st()), "tm_op_adam")
        self._micro = 0
        host = self.p.cpu()
        for k in self.keys:
            self.net.W[k] = host[self.off[k]:self.off[k] + self.net.W[k].numel()].reshape(self.net.W[k].shape).clone()
        self.net._Wd.clear()
        return {"grad_norm": total, "clip_coef": coef}
