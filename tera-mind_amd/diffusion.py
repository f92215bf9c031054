"""Sampler host logic of the path, with the reference's call surface.

`SpacedDiffusionBeatGans.sample(model=, shape=, noise=, pos=, r_start=, imgs=, idx=, patch_size=,
model_kwargs=)` follows reference diffusion/base.py:291-332 (sample), :500-631 (loop driver)
and diffusion/diffusion.py:60-161 (spacing + timestep map).  The integer paths
(`space_timesteps`, `timestep_map`, `sparse_repatch`) and the float64 coefficient tables are
host Python / numpy, exactly as in the reference; the per-pixel update, the pad+patchify
and the UNet run in libteramind_hip.so.

Two keyword-only extensions make fixed-noise parity runs possible (seam S2 of SURVEY.md
section 8b): `x_T=` supplies the initial state of mode A (the reference draws it with
th.randn, base.py:566) and `step_noise=` the per-step DDPM noise (base.py:478).
"""
import ctypes as C
from typing import Callable, List, Optional, Sequence, Union

import numpy as np
import torch

from . import _lib


# ------------------------------------------------------------------------------------------
# integer / float64 host paths
# ------------------------------------------------------------------------------------------
def space_timesteps(num_timesteps: int, section_counts) -> set:
    """Retained timestep set (reference diffusion/diffusion.py:5-57).  "ddimN": the smallest
    integer stride giving exactly N steps; list: per-section `round(k * frac_stride)`
    (Python round = banker's rounding, as in the reference)."""
    if isinstance(section_counts, str):
        if section_counts[:4] in ("ddim", "fdpm"):
            want = int(section_counts[4:])
            for stride in range(1, num_timesteps):
                picked = range(0, num_timesteps, stride)
                if len(picked) == want:
                    return set(picked)
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(v) for v in section_counts.split(",")]
    base, extra = divmod(num_timesteps, len(section_counts))
    out, start = [], 0
    for sec, count in enumerate(section_counts):
        size = base + (1 if sec < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        stride = 1 if count <= 1 else (size - 1) / (count - 1)
        pos = 0.0
        for _ in range(count):
            out.append(start + round(pos))
            pos += stride
        start += size
    return set(out)


def named_beta_schedule(name: str, T: int) -> np.ndarray:
    """reference diffusion/base.py:649-667 ('linear' is the only schedule the path uses)."""
    if name != "linear":
        raise NotImplementedError(f"beta schedule {name!r}")
    scale = 1000 / T
    return np.linspace(scale * 0.0001, scale * 0.02, T, dtype=np.float64)


def sparse_repatch(rna, sz: int):
    """COO coordinate remap image grid -> patch grid (reference diffusion/base.py:111-120).
    Bit-exact integer arithmetic; unlike the reference it does not mutate `crd` in place."""
    dat, crd, ssz = rna
    p1, p2 = ssz[1] // sz, ssz[2] // sz
    crd = crd.long()
    new0 = crd[0] * (p1 * p2) + torch.div(crd[1], sz, rounding_mode="floor") * p2 + torch.div(crd[2], sz, rounding_mode="floor")
    out = torch.stack([new0, crd[1] % sz, crd[2] % sz, crd[3]])
    return dat, out, torch.Size([ssz[0] * p1 * p2, sz, sz, ssz[-1]])


class SpacedDiffusionBeatGans:
    """Tables of GaussianDiffusionBeatGans.__init__ (base.py:64-109) over the re-derived betas
    of SpacedDiffusionBeatGans.__init__ (diffusion.py:76-94); float64 numpy throughout."""

    def __init__(self, T: int, gen_type: str = "ddim", T_train: int = 1000, beta_scheduler: str = "linear"):
        if gen_type not in ("ddpm", "ddim"):
            raise NotImplementedError(gen_type)
        self.gen_type = gen_type
        self.original_num_steps = T_train
        # TrainConfig._make_diffusion_conf (config.py:190-197)
        self.use_timesteps = space_timesteps(T_train, [T] if gen_type == "ddpm" else f"ddim{T}")
        base_betas = named_beta_schedule(beta_scheduler, T_train)
        base_acp = np.cumprod(1.0 - base_betas, axis=0)
        last, new_betas, self.timestep_map = 1.0, [], []
        for i, acp in enumerate(base_acp):
            if i in self.use_timesteps:
                new_betas.append(1 - acp / last)
                last = acp
                self.timestep_map.append(i)
        betas = np.array(new_betas, dtype=np.float64)
        assert (betas > 0).all() and (betas <= 1).all()
        self.betas = self._betas = betas
        self.num_timesteps = int(betas.shape[0])
        alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.alphas_cumprod_next = np.append(self.alphas_cumprod[1:], 0.0)
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = np.log(np.append(self.posterior_variance[1], self.posterior_variance[1:]))
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - self.alphas_cumprod)
        # ModelVarType.fixed_large (base.py:403-413)
        self.model_log_variance = np.log(np.append(self.posterior_variance[1], betas[1:]))

    # ---- per-step scalars handed to the kernel ------------------------------------------------
    def step_coefs(self, i: int) -> _lib.TmStepCoefs:
        """Table entries at index i cast `.float()` (base.py:643); sigma and the two DDIM square
        roots are evaluated in float32 torch exactly like the reference's tensor expressions."""
        f32 = lambda a: torch.tensor(np.float32(a[i]))
        c = _lib.TmStepCoefs()
        c.sqrt_recip_alphas_cumprod = float(f32(self.sqrt_recip_alphas_cumprod))
        c.sqrt_recipm1_alphas_cumprod = float(f32(self.sqrt_recipm1_alphas_cumprod))
        c.posterior_mean_coef1 = float(f32(self.posterior_mean_coef1))
        c.posterior_mean_coef2 = float(f32(self.posterior_mean_coef2))
        c.sigma = float(torch.exp(0.5 * f32(self.model_log_variance))) if i != 0 else 0.0   # nonzero_mask, base.py:476
        ab_prev = f32(self.alphas_cumprod_prev)
        c.sqrt_alpha_bar_prev = float(torch.sqrt(ab_prev))
        c.sqrt_one_minus_alpha_bar_prev = float(torch.sqrt(1 - ab_prev))
        return c

    # ---- training objective, forward half ------------------------------------
    def q_sample(self, x_start: torch.Tensor, t: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
        """q(x_t | x_0) (base.py:141-158): table entries cast .float() and broadcast per image."""
        a = torch.from_numpy(self.sqrt_alphas_cumprod).to(x_start.device)[t].float().reshape(-1, 1, 1, 1)
        b = torch.from_numpy(self.sqrt_one_minus_alphas_cumprod).to(x_start.device)[t].float().reshape(-1, 1, 1, 1)
        return a * x_start + b * noise

    def training_losses(self, model, x_start, r_start, imgs, t, pos, loss_mask, idx=None, patch_size=64,
                        model_kwargs=None, noise=None, *, crop_index=None, loss_type: str = "mse"):
        """GaussianDiffusionBeatGans.training_losses (base.py:181-289): diffuse the padded image, draw a random
        2 x 2-patch window (`crop_index=(ix, iy)` pins the reference's two `random.randrange` draws), crop the COO
        genes and the image to it, run the model once (P = 1: the four original patches -> `pred2`, the shifted
        collage patch -> `pred`) and return {'loss', 'x_t'}: mse (or l1) of both predictions against the noise.
        This is the FORWARD of the training step through the inference kernels -- the value a validation pass or a
        loss curve needs; the model runs as in `.eval()` (the reference's ResBlock dropout p = 0.1, config_parm.py:46, is
        not applied).  The same objective WITH gradients and the optimizer step: train_model.training_loss_and_grads /
        AdamTrainer."""
        import random
        dev = x_start.device
        if noise is None:
            noise = torch.randn_like(x_start)
        halfp = patch_size // 2
        t = t.to(dev).long()
        t_cur = t.repeat_interleave(x_start.shape[0] // t.shape[0])                        # base.py:213-214
        x_t = self.q_sample(x_start, t_cur, noise)
        if loss_mask is not None:
            x_t = x_t * loss_mask
        terms = {"x_t": x_t}
        if crop_index is None:
            crop_index = (random.randrange(pos.shape[0] - 1), random.randrange(pos.shape[1] - 1))   # base.py:220-221
        ix, iy = crop_index
        dat, crd, ssz = r_start
        r_size = patch_size // (x_start.shape[2] // ssz[1])
        crd = crd.long()
        keep = (ix * r_size <= crd[1]) & (crd[1] < (ix + 2) * r_size) & (iy * r_size <= crd[2]) & (crd[2] < (iy + 2) * r_size)
        dat, crd = dat[keep], crd[:, keep].clone()
        crd[1] -= ix * r_size
        crd[2] -= iy * r_size
        rna_pat = sparse_repatch((dat, crd, (ssz[0], 2 * r_size, 2 * r_size, ssz[-1])), r_size)
        sl = (slice(None), slice(None), slice(ix * patch_size, (ix + 2) * patch_size),
              slice(iy * patch_size, (iy + 2) * patch_size))

        def tiles2(a):                                                                     # 'b c (p1 h) (p2 w) -> (b p1 p2) c h w'
            b_, c_, H_, W_ = a.shape
            return a.reshape(b_, c_, 2, patch_size, 2, patch_size).permute(0, 2, 4, 1, 3, 5).reshape(b_ * 4, c_, patch_size, patch_size)

        x_p, n_p, m_p = tiles2(x_t[sl]), tiles2(noise[sl]), tiles2(loss_mask[sl])
        tm = torch.tensor(self.timestep_map, dtype=torch.int64, device=dev)[t]            # _WrappedModel (diffusion.py:140-147)
        shape_only = torch.empty((t.shape[0], x_start.shape[1], patch_size, patch_size), device="meta")   # do_train: p1 = p2 = 2
        out = model(x=x_p, t=tm, rna=rna_pat, imgs=shape_only, patch_size=patch_size, want_pred2=True)
        b_ = t.shape[0]
        n_img = n_p.reshape(b_, 2, 2, -1, patch_size, patch_size).permute(0, 3, 1, 4, 2, 5).reshape(b_, -1, 2 * patch_size, 2 * patch_size)
        noise_shift = n_img[:, :, halfp:-halfp, halfp:-halfp]                              # loss_mask given: the centre patch
        f = (lambda d: d ** 2) if loss_type == "mse" else (lambda d: d.abs())
        flat = lambda a: a.reshape(a.shape[0], -1).mean(1)
        terms["loss"] = flat(f(noise_shift - out.pred)).mean() + flat(f(n_p - out.pred2) * m_p).mean()
        return terms

    # ---- reference-shaped entry point ------------------------------------------------------------
    def sample(self, model, shape=None, noise=None, pos=None, cond=None, x_start=None, r_start=None, imgs=None,
               clip_denoised=True, idx=None, patch_size=64, model_kwargs=None, progress=False, *,
               x_T: Optional[torch.Tensor] = None,
               step_noise: Union[None, Sequence[torch.Tensor], Callable[[int], torch.Tensor]] = None):
        """mode A (idx None): full reverse loop from x_T over an image of `shape`;
        mode B (idx given, `imgs` = padded patch batch): the single step `idx` (test_brn).
        `clip_denoised` must be True (the value both reference callers use, base.py:305; test_brn.py:209-217 and
        experiment.py:325-330 never pass it): the x0 clamp of base.py:423-427 is part of the step kernel.
        `cond` / `x_start` are forwarded to the model by the reference (base.py:311-316) and ignored by the `ours` model;
        they are accepted and ignored here too."""
        if not clip_denoised:
            raise NotImplementedError("clip_denoised=False: tm_sampler_step always clamps the x0 prediction to [-1, 1] "
                                      "(diffusion/base.py:423-427); the reference never samples without it")
        final = None
        for final in self.sample_progressive(model, shape, noise, r_start, imgs, idx, x_T=x_T, step_noise=step_noise):
            pass
        return final

    def sample_progressive(self, model, shapes, noise, rna, img_patch, idx, *, x_T=None, step_noise=None):
        device = model.device if hasattr(model, "device") else next(model.parameters()).device
        if img_patch is not None and idx is None:
            raise AssertionError("imgs given without idx")                     # base.py:564-565
        b, c, H, W = shapes
        # RNG draw order of the reference: the state first (base.py:566) ...
        img = x_T.to(device) if x_T is not None else torch.randn(tuple(shapes), device=device)
        ps = noise.shape[2]                                                   # base.py:568
        P1, P2 = H // ps, W // ps
        indices = list(range(self.num_timesteps))[::-1] if idx is None else [idx]
        rna_msk = None
        precomputed = hasattr(rna, "buf") and hasattr(rna, "p1")            # unet.RnaPyramid / unet.RnaLevel0: handed through
        if torch.is_tensor(rna) or precomputed:
            rna_new = rna
        elif isinstance(rna, (tuple, list)) and len(rna) == 2:
            rna_new, rna_msk = rna
        else:
            r_sz = ps // ((H + ps) // rna[2][1])                              # base.py:594
            rna_new = sparse_repatch(rna, r_sz)
        shape_only = torch.empty((b, c, H, W), device="meta")                 # model reads imgs.shape only
        if len(indices) > 1 and hasattr(model, "precompute_rna") and not precomputed:
            # mode A: the genes are the same in every step of the reverse loop; the reference recomputes get_rna
            # (unet_ours.py:376) per step, here the conditioning pyramid is computed once (bit-identical results)
            rna_new = model.precompute_rna(rna_new, b, imgs=shape_only, patch_size=ps)
        for k, i in enumerate(indices):
            t = torch.full((b,), self.timestep_map[i], dtype=torch.int64, device=device)   # _WrappedModel, diffusion.py:140-147
            x_patches = pad_patchify(img, ps) if img_patch is None else img_patch.to(device).float().contiguous()
            eps = model(x=x_patches, t=t, rna=rna_new, imgs=shape_only, patch_size=ps, idx=idx).pred
            nz = None
            if self.gen_type == "ddpm":
                if img_patch is not None:
                    nz = noise                                                # base.py:617
                elif step_noise is not None:
                    nz = step_noise(k) if callable(step_noise) else step_noise[k]
                else:
                    nz = torch.randn(x_patches.shape, device=device)          # base.py:478
                nz = nz.to(device).float().contiguous()
            img = sampler_step(self, i, x_patches, eps, nz, b, P1, P2)
            if rna_msk is not None:
                img = img * rna_msk + rna_msk - 1                             # base.py:629-630
            yield img


# ------------------------------------------------------------------------------------------
# thin wrappers of the C-ABI kernels
# ------------------------------------------------------------------------------------------
def gen_sample(model, sampler: "SpacedDiffusionBeatGans", pnm_w: int, pnm_h: int, x_start: torch.Tensor, r_start, patch_size: int,
               sample_size: int, *, x_T: Optional[torch.Tensor] = None, start: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The sampling half of `LitModel.gen_sample` (experiment.py:293-330; the picture grid / TensorBoard part behind it is
    not part of the path): the patch-shaped noise tensor that fixes the batch (`sample_size * pnm_w * pnm_h` patches ->
    `bat` images), the first `bat` images' entries of the COO gene triple `(dat, crd, ssz)` (:311-318), the patch-corner
    grid `pos` (:303-308, unused by the `ours` model) and ONE `sampler.sample` call in mode A (:325-330).
    `x_T` replaces the reference's `torch.randn` patch tensor (only its batch and patch size matter to the sampler);
    `start` pins the initial state the sampler would draw itself (base.py:566)."""
    dev = model.device if hasattr(model, "device") else next(model.parameters()).device
    if x_T is None:
        x_T = torch.randn(sample_size * pnm_w * pnm_h, x_start.shape[1], patch_size, patch_size, device=dev)
    bat = len(x_T) // pnm_w // pnm_h
    xs = x_start[:bat]
    gx = torch.linspace(0, pnm_w, pnm_w + 1, device=dev)
    gy = torch.linspace(0, pnm_h, pnm_h + 1, device=dev)
    xx, yy = torch.meshgrid(gx, gy, indexing="ij")
    pos = torch.stack([xx, yy], dim=-1).flatten(0, 1).repeat(xs.shape[0], 1)
    dat, crd, ssz = r_start
    crd = crd.long()
    keep = crd[0] < bat
    rs = (dat[keep], crd[:, keep], torch.Size([bat, ssz[1], ssz[2], ssz[3]]))
    return sampler.sample(model=model, shape=xs.shape, noise=x_T.detach(), r_start=rs, patch_size=patch_size, pos=pos, x_T=start)


def pad_patchify(img: torch.Tensor, ps: int, pad_value: float = 0.0) -> torch.Tensor:
    """F.pad(img, ps/2) + 'b c (p1 h) (p2 w) -> (b p1 p2) c h w' (base.py:606-607)."""
    img = img.float().contiguous()
    b, c, H, W = img.shape
    P1, P2 = H // ps, W // ps
    out = torch.empty((b * (P1 + 1) * (P2 + 1), c, ps, ps), dtype=torch.float32, device=img.device)
    with torch.cuda.device(img.device):
        _lib.check(_lib.lib().tm_pad_patchify(_lib.ptr(img), _lib.ptr(out), b, c, P1, P2, ps, pad_value,
                                              _lib.current_stream_ptr()), "tm_pad_patchify")
    return out


def sampler_step(smp: SpacedDiffusionBeatGans, i: int, x_patches, eps, noise, b: int, P1: int, P2: int) -> torch.Tensor:
    x_patches, eps = x_patches.float().contiguous(), eps.float().contiguous()
    n, c, ps, _ = x_patches.shape
    assert n == b * (P1 + 1) * (P2 + 1) and eps.shape[0] == b * P1 * P2
    out = torch.empty((b, c, P1 * ps, P2 * ps), dtype=torch.float32, device=x_patches.device)
    coefs = smp.step_coefs(i)
    mode = 0 if smp.gen_type == "ddpm" else 1
    nz = noise if (mode == 0 and i != 0) else None
    with torch.cuda.device(x_patches.device):
        _lib.check(_lib.lib().tm_sampler_step(C.byref(coefs), _lib.ptr(x_patches), _lib.ptr(eps), _lib.ptr(nz), _lib.ptr(out),
                                              b, P1, P2, c, ps, mode, _lib.current_stream_ptr()), "tm_sampler_step")
    return out
