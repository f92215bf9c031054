"""Weight contract of the path: the reference checkpoint's key set, and a deterministic
generator used wherever real checkpoints are unavailable (tests, bench, smoke).

`param_spec(cfg)` enumerates `(key, shape)` exactly as `model.state_dict()` of the
reference lists them (reference: model/unet_ours.py:82-296 constructor wiring; verified
against tests/golden/state_dict_keys_*.json which was dumped from the reference).

`hashed_state_dict(cfg, seed)` fills every tensor from an integer hash of
(key, flat index): no tensor is left at its init value -- in particular the
`zero_module` convs (reference model/MBAblocks.py:187-188) are overwritten, otherwise a
"random init" run would exercise only half of each ResBlock.
"""
import zlib
from typing import Dict, List, Tuple

import numpy as np

from .config import PathConfig


def _resblock(keys, pfx, cin, cout, emb):
    keys.append((f"{pfx}.in_layers.0.weight", (1, cin, 1, 1)))
    keys.append((f"{pfx}.in_layers.2.weight", (cout, cin, 3, 3, 3)))
    keys.append((f"{pfx}.in_layers.2.bias", (cout,)))
    keys.append((f"{pfx}.emb_layers.1.weight", (2 * cout, emb)))
    keys.append((f"{pfx}.emb_layers.1.bias", (2 * cout,)))
    keys.append((f"{pfx}.out_layers.0.weight", (1, cout, 1, 1)))
    keys.append((f"{pfx}.out_layers.3.weight", (cout, cout, 3, 3, 3)))
    keys.append((f"{pfx}.out_layers.3.bias", (cout,)))
    if cin != cout:
        keys.append((f"{pfx}.skip_connection.weight", (cout, cin, 1, 1, 1)))
        keys.append((f"{pfx}.skip_connection.bias", (cout,)))


def _attnblock(keys, pfx, c, g):
    keys.append((f"{pfx}.norm1.weight", (c,)))
    for n in ("q", "k", "v"):
        keys.append((f"{pfx}.attn.{n}.weight", (c, c)))
        keys.append((f"{pfx}.attn.{n}.bias", (c,)))
    keys.append((f"{pfx}.attn.q_norm.weight", (c,)))
    keys.append((f"{pfx}.attn.k_norm.weight", (c,)))
    keys.append((f"{pfx}.attn.proj.weight", (c, c)))
    keys.append((f"{pfx}.attn.proj.bias", (c,)))
    keys.append((f"{pfx}.norm2.weight", (c,)))
    keys.append((f"{pfx}.mlp.fc1.weight", (4 * c, c)))
    keys.append((f"{pfx}.mlp.fc1.bias", (4 * c,)))
    keys.append((f"{pfx}.mlp.fc2.weight", (c, 4 * c)))
    keys.append((f"{pfx}.mlp.fc2.bias", (c,)))
    keys.append((f"{pfx}.adaLN_modulation.1.weight", (7 * c, g)))
    keys.append((f"{pfx}.adaLN_modulation.1.bias", (7 * c,)))


def _gene_block(keys, cfg: PathConfig):
    d, g, kz = cfg.gene_hidden, cfg.rna_num, cfg.down_z_kernel
    p = "rna_blocks.0.0"
    for n in ("q", "v"):
        keys.append((f"{p}.attn.{n}.weight", (d, d)))
        keys.append((f"{p}.attn.{n}.bias", (d,)))
    keys.append((f"{p}.attn.q_norm.weight", (d,)))
    keys.append((f"{p}.attn.proj.weight", (d, d)))
    keys.append((f"{p}.attn.proj.bias", (d,)))
    keys.append((f"{p}.norm2.weight", (d,)))
    keys.append((f"{p}.mlp.fc1.weight", (4 * d, d)))
    keys.append((f"{p}.mlp.fc1.bias", (4 * d,)))
    keys.append((f"{p}.mlp.fc2.weight", (d, 4 * d)))
    keys.append((f"{p}.mlp.fc2.bias", (d,)))
    keys.append((f"{p}.down_z.weight", (g, g, kz, 3, 3)))
    keys.append((f"{p}.down_z.bias", (g,)))


def param_spec(cfg: PathConfig, vis_only: bool = False) -> List[Tuple[str, Tuple[int, ...]]]:
    """[(reference state_dict key, shape)] in the reference's registration order.
    `vis_only` = the attention-map model (reference model/unet_attn.py: time_embed +
    rna_blocks[0] only)."""
    keys: List[Tuple[str, Tuple[int, ...]]] = []
    E, ch0 = cfg.embed_ch, cfg.net_ch
    keys += [("time_embed.time_embed.0.weight", (E, ch0)), ("time_embed.time_embed.0.bias", (E,)),
             ("time_embed.time_embed.2.weight", (E, E)), ("time_embed.time_embed.2.bias", (E,))]
    _gene_block(keys, cfg)
    if vis_only:
        return keys
    rw = cfg.rna_widths
    ich = (rw[0],) + rw[:-1]
    for rid in range(1, 4):
        keys.append((f"rna_blocks.{rid}.1.weight", (rw[rid], ich[rid], 1, 3, 3)))
        keys.append((f"rna_blocks.{rid}.1.bias", (rw[rid],)))

    n_stain_ch = cfg.in_channels // cfg.z_size
    keys.append(("input_blocks.0.0.weight", (ch0, n_stain_ch, 1, 3, 3)))
    keys.append(("input_blocks.0.0.bias", (ch0,)))
    L = len(cfg.ch_mult)
    ch, res, k = ch0, cfg.patch_size, 1
    enc_ch = [[] for _ in range(L)]
    enc_ch[0].append(ch)
    for lvl, mult in enumerate(cfg.ch_mult):
        rd = rw[L - 1 - lvl]
        for _ in range(cfg.num_res_blocks):
            cout = mult * ch0
            _resblock(keys, f"input_blocks.{k}.0", ch + rd, cout, E)
            ch = cout
            if res in cfg.attn_res:
                _attnblock(keys, f"input_blocks.{k}.1", ch, rd)
            enc_ch[lvl].append(ch)
            k += 1
        if lvl != L - 1:
            res //= 2
            _resblock(keys, f"input_blocks.{k}.0", ch, ch, E)
            enc_ch[lvl + 1].append(ch)
            k += 1
    _resblock(keys, "middle_block.0", ch + rw[0], ch, E)
    _attnblock(keys, "middle_block.1", ch, rw[0])
    _resblock(keys, "middle_block.2", ch, ch, E)
    k = 0
    for lvl in reversed(range(L)):
        rd = rw[L - 1 - lvl]
        for i in range(cfg.num_res_blocks + 1):
            skip = enc_ch[lvl].pop()
            cout = cfg.ch_mult[lvl] * ch0
            _resblock(keys, f"output_blocks.{k}.0", ch + skip + rd, cout, E)
            ch = cout
            nxt = 1
            if res in cfg.attn_res:
                _attnblock(keys, f"output_blocks.{k}.1", ch, rd)
                nxt = 2
            if lvl and i == cfg.num_res_blocks:
                res *= 2
                _resblock(keys, f"output_blocks.{k}.{nxt}", ch, ch, E)
            k += 1
    keys.append(("out.0.weight", (1, ch, 1, 1)))
    keys.append(("out.2.weight", (n_stain_ch, ch0, 1, 3, 3)))
    keys.append(("out.2.bias", (n_stain_ch,)))
    return keys


# ------------------------------------------------------------------------------------------
# integer-hash generator
# ------------------------------------------------------------------------------------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wraps mod 2^64)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return x ^ (x >> np.uint64(31))


def hashed_uniform(key: str, n: int, seed: int = 0) -> np.ndarray:
    """n float64 values uniform in [-1, 1), a pure function of (key, seed, index)."""
    base = np.uint64((zlib.crc32(key.encode()) << 32) ^ (seed & 0xFFFFFFFF))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + base * np.uint64(0x2545F4914F6CDD1D)
    h = _splitmix64(idx)
    return (h >> np.uint64(11)).astype(np.float64) * (2.0 / 9007199254740992.0) - 1.0


def _std_for(key: str, shape) -> Tuple[float, float]:
    """(mean, std) per tensor family: keeps activations O(1) through ~50 layers so that
    parity differences are not hidden by tiny or exploding magnitudes."""
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "bias":
        return 0.0, 0.05
    norm_like = (len(shape) == 1 or (len(shape) == 4 and shape[0] == 1))
    if norm_like:                      # RMSNorm weights: init 1 (MBAblocks.py:29-31)
        return 1.0, 0.2
    fan_in = int(np.prod(shape[1:]))
    gain = 1.0
    if "out_layers.3" in key:          # second conv of a ResBlock (zero-init upstream)
        gain = 0.5
    if "adaLN_modulation" in key:
        gain = 0.5
    return 0.0, gain / np.sqrt(fan_in)


def hashed_tensor(key: str, shape, seed: int = 0) -> np.ndarray:
    n = int(np.prod(shape))
    mean, std = _std_for(key, tuple(shape))
    u = hashed_uniform(key, n, seed)
    return (mean + (np.sqrt(3.0) * std) * u).astype(np.float32).reshape(shape)


def hashed_state_dict(cfg: PathConfig, seed: int = 0, vis_only: bool = False) -> Dict[str, "object"]:
    """{key: torch.FloatTensor} for every parameter of the path (reference key names)."""
    import torch
    return {k: torch.from_numpy(hashed_tensor(k, s, seed)) for k, s in param_spec(cfg, vis_only)}


def strip_lightning_state_dict(state: dict) -> dict:
    """Checkpoint import contract (reference test_brn.py:140-147): take `state_dict`,
    drop `*ema_model*`, strip the `model.` prefix."""
    sd = state["state_dict"] if "state_dict" in state else state
    out = {}
    for key, val in sd.items():
        if "ema_model" in key:
            continue
        out[key.replace("model.", "")] = val
    return out
