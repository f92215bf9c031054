"""Training slice: one ResBlock's training-mode forward and backward on the HIP kernels.

The reference trains through torch.autograd over stock modules (experiment.py:121-193 -> diffusion/base.py:181-289 ->
model/MBAblocks.py:237-299).  This module is the first slice of a native training path: `ResBlockTrain` runs

    A   = SiLU(RMSNorm(x) * w1)                                   in_layers[0:2]          (MBAblocks.py:141-149)
    H1  = Conv3d_3x3x3(A) + b1                                    in_layers[2]
    D   = Dropout(SiLU(RMSNorm(H1) * w2 * (1 + scale) + shift))   out_layers[0:3] + apply_conditions (:196-203, :356-367)
    out = skip(x) + Conv3d_3x3x3(D) + b2                          out_layers[3], skip_connection (:220-224, :297)

and its backward with hand-written kernels: the conv data gradients on the forward MFMA conv kernel (flipped, transposed
weights), `conv_wgrad_kernel`, `prep_bwd_kernel`, `chan_sum_kernel` (csrc/tm_train.hip).  The dropout keep mask is an
INPUT (the reference draws it inside nn.Dropout(p=0.1), config_parm.py:46), so that gradients can be compared with
torch.autograd of the CPU oracle on the same mask.  `AttnBlockTrain` is the gene cross-attention block (MBAblocks.py:428-514)
forward and backward, checked against the reference module's own autograd.  The whole-model training step (both decoder
passes, the gene-gene attention block, time embedding, loss, clip + Adam) is composed in train_model.py.  Not covered:
mixed precision (the reference trains under fp16 autocast; this slice is fp32), EMA (commented out upstream,
experiment.py:200), the data loader -- see DESIGN.md section 8.
"""
import ctypes as C
from typing import Dict, Optional

import torch

from . import _lib


def _cb8(x: torch.Tensor) -> torch.Tensor:
    N, Cc, Z, H, W = x.shape
    y = torch.empty((N, (Cc + 7) // 8, Z, H, W, 8), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().tm_op_to_cb8(_lib.ptr(x.contiguous()), _lib.ptr(y), N, Cc, Z, H, W, _lib.current_stream_ptr()))
    return y


def _ncdhw(y: torch.Tensor, Cc: int) -> torch.Tensor:
    N, cb, Z, H, W, _ = y.shape
    x = torch.empty((N, Cc, Z, H, W), dtype=torch.float32, device=y.device)
    _lib.check(_lib.lib().tm_op_from_cb8(_lib.ptr(y), _lib.ptr(x), N, Cc, Z, H, W, _lib.current_stream_ptr()))
    return x


def _host(t: Optional[torch.Tensor]):
    return None if t is None else t.detach().to("cpu", torch.float32).contiguous()


def _hp(t):
    return C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())


class ResBlockTrain:
    """One ResBlock (Z = 2, 'same' resolution) with parameters given as a dict with the reference's key suffixes:
    in_layers.0.weight [1,C,1,1], in_layers.2.{weight,bias}, out_layers.0.weight, out_layers.3.{weight,bias},
    skip_connection.{weight,bias} (when Cin != Cout).  forward(x, scale, shift, keep_mask, p) -> out; backward(dout) ->
    (dx, dscale, dshift, grads) with grads keyed like the parameters."""

    def __init__(self, params: Dict[str, torch.Tensor], device="cuda:0"):
        self.p = {k: _host(v) for k, v in params.items()}
        self.dev = torch.device(device)
        self.cin = self.p["in_layers.2.weight"].shape[1]
        self.cout = self.p["in_layers.2.weight"].shape[0]
        self.has_skip = "skip_connection.weight" in self.p
        self._saved = None

    # -- pieces ---------------------------------------------------------------------------------
    def _prep(self, x_cb, nw, scale, shift, mask_cb, drop_scale, per_image, N, Cc, Z, S):
        y = torch.empty_like(x_cb)
        _lib.check(_lib.lib().tm_op_prep_train(_lib.ptr(x_cb), _hp(nw), _hp(scale), _hp(shift), _lib.ptr(mask_cb), drop_scale, per_image,
                                               _lib.ptr(y), N, Cc, Z, S, _lib.current_stream_ptr()), "tm_op_prep_train")
        return y

    def _conv(self, x_cb, w, b, N, Cin, Cout, Z, S, ksize):
        y = torch.zeros((N, (Cout + 7) // 8, Z, S, S, 8), dtype=torch.float32, device=self.dev)
        _lib.check(_lib.lib().tm_op_conv_mfma(_lib.ptr(x_cb), _hp(w), _hp(b), _lib.ptr(y), N, Cin, Cout, Z, S, ksize, 0, 0, 0,
                                              _lib.current_stream_ptr()), "tm_op_conv_mfma")
        return y

    def forward(self, x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, keep_mask: Optional[torch.Tensor] = None,
                p_drop: float = 0.0, per_image: int = 1) -> torch.Tensor:
        """x [N, Cin, 2, S, S] (device); scale / shift [ceil(N / per_image), Cout]; keep_mask [N, Cout, 2, S, S] of 0 / 1."""
        N, Cin, Z, S, _ = x.shape
        assert Cin == self.cin and Z == 2
        P = self.p
        x_cb = _cb8(x.to(self.dev))
        w1n, w2n = P["in_layers.0.weight"].reshape(-1), P["out_layers.0.weight"].reshape(-1)
        sc, sh = _host(scale), _host(shift)
        mask_cb = _cb8(keep_mask.to(self.dev).float()) if keep_mask is not None else None
        ds = 1.0 / (1.0 - p_drop) if keep_mask is not None else 1.0
        A = self._prep(x_cb, w1n, None, None, None, 1.0, per_image, N, Cin, Z, S)
        H1 = self._conv(A, P["in_layers.2.weight"], P["in_layers.2.bias"], N, Cin, self.cout, Z, S, 3)
        D = self._prep(H1, w2n, sc, sh, mask_cb, ds, per_image, N, self.cout, Z, S)
        H2 = self._conv(D, P["out_layers.3.weight"], P["out_layers.3.bias"], N, self.cout, self.cout, Z, S, 3)
        if self.has_skip:
            sk = self._conv(x_cb, P["skip_connection.weight"], P["skip_connection.bias"], N, Cin, self.cout, Z, S, 1)
        else:
            sk = x_cb
        out_cb = sk + H2                                          # the residual add (elementwise, torch on the device)
        self._saved = dict(x_cb=x_cb, A=A, H1=H1, D=D, mask_cb=mask_cb, ds=ds, sc=sc, sh=sh, per_image=per_image, shape=(N, Z, S))
        return _ncdhw(out_cb, self.cout)

    def _dgrad(self, dy_cb, w, N, Cin, Cout, Z, S, ksize):
        dx = torch.zeros((N, (Cin + 7) // 8, Z, S, S, 8), dtype=torch.float32, device=self.dev)
        _lib.check(_lib.lib().tm_op_conv_dgrad(_lib.ptr(dy_cb), _hp(w), _lib.ptr(dx), N, Cin, Cout, Z, S, ksize,
                                               _lib.current_stream_ptr()), "tm_op_conv_dgrad")
        return dx

    def _wgrad(self, x_cb, dy_cb, N, Cin, Cout, Z, S, ksize):
        taps = 27 if ksize == 3 else 1
        dw = torch.empty((Cout, Cin) + ((3, 3, 3) if ksize == 3 else (1, 1, 1)), dtype=torch.float32)
        db = torch.empty((Cout,), dtype=torch.float32)
        assert dw.numel() == Cout * Cin * taps
        _lib.check(_lib.lib().tm_op_conv_wgrad(_lib.ptr(x_cb), _lib.ptr(dy_cb), _hp(dw), _hp(db), N, Cin, Cout, Z, S, ksize,
                                               _lib.current_stream_ptr()), "tm_op_conv_wgrad")
        return dw, db

    def _prep_bwd(self, x_cb, g_cb, nw, sc, sh, mask_cb, ds, per_image, N, Cc, Z, S):
        dx = torch.empty_like(x_cb)
        nimg = (N + per_image - 1) // per_image
        dw = torch.empty((Cc,), dtype=torch.float32)
        dsc = torch.empty((nimg, Cc), dtype=torch.float32) if sc is not None else None
        dsh = torch.empty((nimg, Cc), dtype=torch.float32) if sc is not None else None
        _lib.check(_lib.lib().tm_op_prep_bwd(_lib.ptr(x_cb), _lib.ptr(g_cb), _hp(nw), _hp(sc), _hp(sh), _lib.ptr(mask_cb), ds, per_image,
                                             _lib.ptr(dx), _hp(dw), _hp(dsc), _hp(dsh), N, Cc, Z, S, _lib.current_stream_ptr()),
                   "tm_op_prep_bwd")
        return dx, dw, dsc, dsh

    def backward(self, dout: torch.Tensor):
        s, P = self._saved, self.p
        N, Z, S = s["shape"]
        Cin, Cout = self.cin, self.cout
        g = _cb8(dout.to(self.dev).float())
        grads = {}
        # out = skip(x) + conv2(D)
        dD = self._dgrad(g, P["out_layers.3.weight"], N, Cout, Cout, Z, S, 3)
        grads["out_layers.3.weight"], grads["out_layers.3.bias"] = self._wgrad(s["D"], g, N, Cout, Cout, Z, S, 3)
        dH1, dw2, dscale, dshift = self._prep_bwd(s["H1"], dD, P["out_layers.0.weight"].reshape(-1), s["sc"], s["sh"], s["mask_cb"],
                                                  s["ds"], s["per_image"], N, Cout, Z, S)
        grads["out_layers.0.weight"] = dw2.reshape(P["out_layers.0.weight"].shape)
        dA = self._dgrad(dH1, P["in_layers.2.weight"], N, Cin, Cout, Z, S, 3)
        grads["in_layers.2.weight"], grads["in_layers.2.bias"] = self._wgrad(s["A"], dH1, N, Cin, Cout, Z, S, 3)
        dx, dw1, _, _ = self._prep_bwd(s["x_cb"], dA, P["in_layers.0.weight"].reshape(-1), None, None, None, 1.0, s["per_image"], N, Cin, Z, S)
        grads["in_layers.0.weight"] = dw1.reshape(P["in_layers.0.weight"].shape)
        if self.has_skip:
            dx = dx + self._dgrad(g, P["skip_connection.weight"], N, Cin, Cout, Z, S, 1)
            grads["skip_connection.weight"], grads["skip_connection.bias"] = self._wgrad(s["x_cb"], g, N, Cin, Cout, Z, S, 1)
        else:
            dx = dx + g
        return _ncdhw(dx, Cin), dscale, dshift, grads


class AttnBlockTrain:
    """One AttnBlock with gene cross-attention (model/MBAblocks.py:428-514: gene_trans=True, cond given, num_heads 1, n_h 2)
    forward and backward on the HIP kernels.  Parameters as a dict with the reference's key suffixes: norm1.weight,
    norm2.weight, attn.{q,k,v,proj}.{weight,bias}, attn.{q_norm,k_norm}.weight, mlp.{fc1,fc2}.{weight,bias},
    adaLN_modulation.1.{weight,bias}.  forward(x [N,C,Z,S,S], cond [N,G,Z,S,S]) -> out; backward(dout) -> (dx, dcond, grads).

        sc = SiLU(cond);  m = Linear_{G -> 7C}(sc);  shift_msa, scale_msa, gate_msa, crss, shift_mlp, scale_mlp, gate_mlp = m.chunk(7)
        x1 = x + gate_msa * proj(core(q(modulate(norm1, x, shift_msa, scale_msa)), k(crss), v(crss)))
        out = x1 + gate_mlp * fc2(GELU_tanh(fc1(modulate(norm2, x1, shift_mlp, scale_mlp))))

    Every Linear runs as a 1x1x1 conv on the MFMA conv kernel (forward, data gradient with transposed weights) and
    conv_wgrad_kernel; modulate(norm) on prep_kernel / modnorm_bwd_kernel; the windowed attention core on attn_train_kernel;
    the gates and activations on ew_kernel (csrc/tm_train.hip).  Channel chunks / concatenations of the CB8 tensors are torch
    slices on the device (C a multiple of 8)."""

    LIN = ("attn.q", "attn.k", "attn.v", "attn.proj", "mlp.fc1", "mlp.fc2", "adaLN_modulation.1")

    def __init__(self, params: Dict[str, torch.Tensor], device="cuda:0"):
        self.p = {k: _host(v) for k, v in params.items()}
        self.dev = torch.device(device)
        self.C = self.p["attn.q.weight"].shape[0]
        self.G = self.p["adaLN_modulation.1.weight"].shape[1]
        self.hid = self.p["mlp.fc1.weight"].shape[0]
        if self.C % 8:
            raise ValueError("AttnBlockTrain: hidden size must be a multiple of 8")
        self._saved = None

    # -- pieces ---------------------------------------------------------------------------------
    def _geo(self):
        return self._saved["shape"]

    def _lin(self, x_cb, name, cin, cout):
        N, Z, S = self._geo()
        y = torch.zeros((N, (cout + 7) // 8, Z, S, S, 8), dtype=torch.float32, device=self.dev)
        _lib.check(_lib.lib().tm_op_conv_mfma(_lib.ptr(x_cb), _hp(self.p[name + ".weight"]), _hp(self.p[name + ".bias"]), _lib.ptr(y), N, cin,
                                              cout, Z, S, 1, 0, 0, 0, _lib.current_stream_ptr()), "tm_op_conv_mfma")
        return y

    def _lin_bwd(self, x_cb, dy_cb, name, cin, cout, grads, need_dx=True):
        N, Z, S = self._geo()
        dw = torch.empty((cout, cin), dtype=torch.float32)
        db = torch.empty((cout,), dtype=torch.float32)
        _lib.check(_lib.lib().tm_op_conv_wgrad(_lib.ptr(x_cb), _lib.ptr(dy_cb), _hp(dw), _hp(db), N, cin, cout, Z, S, 1,
                                               _lib.current_stream_ptr()), "tm_op_conv_wgrad")
        grads[name + ".weight"], grads[name + ".bias"] = dw, db
        if not need_dx:
            return None
        dx = torch.zeros((N, (cin + 7) // 8, Z, S, S, 8), dtype=torch.float32, device=self.dev)
        _lib.check(_lib.lib().tm_op_conv_dgrad(_lib.ptr(dy_cb), _hp(self.p[name + ".weight"]), _lib.ptr(dx), N, cin, cout, Z, S, 1,
                                               _lib.current_stream_ptr()), "tm_op_conv_dgrad")
        return dx

    def _ew(self, op, a, b=None, c=None, two=False):
        o1 = torch.empty_like(a)
        o2 = torch.empty_like(a) if two else None
        _lib.check(_lib.lib().tm_op_ew(op, _lib.ptr(a), _lib.ptr(b), _lib.ptr(c), _lib.ptr(o1), _lib.ptr(o2), a.numel(),
                                       _lib.current_stream_ptr()), "tm_op_ew")
        return (o1, o2) if two else o1

    def _modnorm(self, x_cb, w, scale, shift):
        N, Z, S = self._geo()
        y = torch.empty_like(x_cb)
        _lib.check(_lib.lib().tm_op_modnorm(_lib.ptr(x_cb), _hp(w), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(y), N, self.C, Z, S,
                                            _lib.current_stream_ptr()), "tm_op_modnorm")
        return y

    def _modnorm_bwd(self, x_cb, g_cb, w, scale):
        N, Z, S = self._geo()
        dx, dsc, dsh = torch.empty_like(x_cb), torch.empty_like(x_cb), torch.empty_like(x_cb)
        dw = torch.empty((self.C,), dtype=torch.float32)
        _lib.check(_lib.lib().tm_op_modnorm_bwd(_lib.ptr(x_cb), _lib.ptr(g_cb), _hp(w), _lib.ptr(scale), _lib.ptr(dx), _lib.ptr(dsc),
                                                _lib.ptr(dsh), _hp(dw), N, self.C, Z, S, _lib.current_stream_ptr()), "tm_op_modnorm_bwd")
        return dx, dsc, dsh, dw

    def _core(self, q, k, v, dout=None):
        N, Z, S = self._geo()
        qw, kw = self.p["attn.q_norm.weight"], self.p["attn.k_norm.weight"]
        if dout is None:
            o = torch.zeros_like(q)
            _lib.check(_lib.lib().tm_op_window_attn_train(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _hp(qw), _hp(kw), None, _lib.ptr(o), None,
                                                          None, None, None, None, N, self.C, Z, S, _lib.current_stream_ptr()),
                       "tm_op_window_attn_train")
            return o
        dq, dk, dv = torch.zeros_like(q), torch.zeros_like(q), torch.zeros_like(q)
        dqw, dkw = torch.empty((self.C,), dtype=torch.float32), torch.empty((self.C,), dtype=torch.float32)
        _lib.check(_lib.lib().tm_op_window_attn_train(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _hp(qw), _hp(kw), _lib.ptr(dout), None,
                                                      _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), _hp(dqw), _hp(dkw), N, self.C, Z, S,
                                                      _lib.current_stream_ptr()), "tm_op_window_attn_train")
        return dq, dk, dv, dqw, dkw

    # -- forward / backward ---------------------------------------------------------------------
    def forward(self, x: torch.Tensor, cond: torch.Tensor) -> torch.Tensor:
        N, Cc, Z, S, _ = x.shape
        assert Cc == self.C and cond.shape[1] == self.G and cond.shape[0] == N and tuple(cond.shape[2:]) == tuple(x.shape[2:])
        return _ncdhw(self.forward_cb(_cb8(x.to(self.dev).float()), _cb8(cond.to(self.dev).float())), self.C)

    def backward(self, dout: torch.Tensor):
        dx, dcond, grads = self.backward_cb(_cb8(dout.to(self.dev).float()))
        return _ncdhw(dx, self.C), _ncdhw(dcond, self.G), grads

    def forward_cb(self, x_cb: torch.Tensor, cond_cb: torch.Tensor) -> torch.Tensor:
        """CB8 in, CB8 out: x_cb [N, C/8, Z, S, S, 8], cond_cb [N, ceil(G/8), Z, S, S, 8]."""
        N, _, Z, S, _, _ = x_cb.shape
        P, Cb = self.p, self.C // 8
        self._saved = dict(shape=(N, Z, S))
        sc = self._ew(4, cond_cb)
        m = self._lin(sc, "adaLN_modulation.1", self.G, 7 * self.C)
        shift_msa, scale_msa, gate_msa, crss, shift_mlp, scale_mlp, gate_mlp = (m[:, i * Cb:(i + 1) * Cb].contiguous() for i in range(7))
        n1 = self._modnorm(x_cb, P["norm1.weight"], scale_msa, shift_msa)
        q = self._lin(n1, "attn.q", self.C, self.C)
        k = self._lin(crss, "attn.k", self.C, self.C)
        v = self._lin(crss, "attn.v", self.C, self.C)
        o = self._core(q, k, v)
        pr = self._lin(o, "attn.proj", self.C, self.C)
        x1 = self._ew(0, x_cb, gate_msa, pr)
        n2 = self._modnorm(x1, P["norm2.weight"], scale_mlp, shift_mlp)
        h = self._lin(n2, "mlp.fc1", self.C, self.hid)
        a = self._ew(2, h)
        f = self._lin(a, "mlp.fc2", self.hid, self.C)
        out = self._ew(0, x1, gate_mlp, f)
        self._saved.update(x_cb=x_cb, cond_cb=cond_cb, sc=sc, scale_msa=scale_msa, gate_msa=gate_msa, crss=crss, scale_mlp=scale_mlp,
                           gate_mlp=gate_mlp, n1=n1, q=q, k=k, v=v, o=o, pr=pr, x1=x1, n2=n2, h=h, a=a, f=f)
        return out

    def backward_cb(self, g: torch.Tensor):
        s, P = self._saved, self.p
        C_, G, hid = self.C, self.G, self.hid
        grads: Dict[str, torch.Tensor] = {}
        # out = x1 + gate_mlp * f
        d_f, d_gate_mlp = self._ew(1, g, s["gate_mlp"], s["f"], two=True)
        d_a = self._lin_bwd(s["a"], d_f, "mlp.fc2", hid, C_, grads)
        d_h = self._ew(3, d_a, s["h"])
        d_n2 = self._lin_bwd(s["n2"], d_h, "mlp.fc1", C_, hid, grads)
        dx1b, dscale_mlp, dshift_mlp, grads["norm2.weight"] = self._modnorm_bwd(s["x1"], d_n2, P["norm2.weight"], s["scale_mlp"])
        dx1 = self._ew(6, g, dx1b)
        # x1 = x + gate_msa * pr
        d_pr, d_gate_msa = self._ew(1, dx1, s["gate_msa"], s["pr"], two=True)
        d_o = self._lin_bwd(s["o"], d_pr, "attn.proj", C_, C_, grads)
        dq, dk, dv, grads["attn.q_norm.weight"], grads["attn.k_norm.weight"] = self._core(s["q"], s["k"], s["v"], d_o)
        d_n1 = self._lin_bwd(s["n1"], dq, "attn.q", C_, C_, grads)
        d_crss = self._ew(6, self._lin_bwd(s["crss"], dk, "attn.k", C_, C_, grads), self._lin_bwd(s["crss"], dv, "attn.v", C_, C_, grads))
        dxa, dscale_msa, dshift_msa, grads["norm1.weight"] = self._modnorm_bwd(s["x_cb"], d_n1, P["norm1.weight"], s["scale_msa"])
        dx = self._ew(6, dx1, dxa)
        dm = torch.cat([dshift_msa, dscale_msa, d_gate_msa, d_crss, dshift_mlp, dscale_mlp, d_gate_mlp], dim=1).contiguous()
        d_sc = self._lin_bwd(s["sc"], dm, "adaLN_modulation.1", G, 7 * C_, grads)
        dcond = self._ew(5, d_sc, s["cond_cb"])
        return dx, dcond, grads
