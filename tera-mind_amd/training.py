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
torch.autograd of the CPU oracle on the same mask.  Not covered yet: the AttnBlock / gene-attention backward, the optimizer,
EMA, mixed precision -- see DESIGN.md section 8.
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
