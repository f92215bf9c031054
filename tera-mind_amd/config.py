"""Config surface of the denoising hot path.

Restates the twelve fields of the reference config that the path needs
(reference: config_parm.py:5-59 `prep_config_parm`, config.py:293-325 `make_model_conf`,
test_brn.py:337-344 checkpoint-directory-name parsing).  Everything else in the
reference's ~140-field TrainConfig is training / legacy and out of scope.
"""
import math
from dataclasses import dataclass, field
from typing import Tuple

# reference: model/unet_ours.py:278-279 -- RNA pyramid widths are hard-coded there.
RNA_PYRAMID_TAIL = (128, 64, 32)
# reference: model/MBAblocks.py:472 -- down_z kernel depth per rna_slc.
DOWN_Z_KERNEL = {1: 1, 4: 3, 8: 5, 16: 9}
# reference: utils/MBADataset_tst.py:30 -- z padding (in gene slices) per rna_slc.
Z_PAD = {1: 0, 4: 1, 8: 1, 16: 3}
GENES_PER_SLICE = 500      # reference: model/unet_ours.py:308-309 ('(z g)', g=500)


@dataclass
class PathConfig:
    """`patch_size` / `rna_slc` / `stain` / `rna_num` are the reference's CLI surface
    (train.py:9-39); the rest are the fixed values prep_config_parm sets."""
    patch_size: int = 64
    rna_slc: int = 4
    stain: str = "all"
    rna_num: int = 229
    mouse: str = "638850"
    method: str = "ours"
    net_ch: int = 64
    ch_mult: Tuple[int, ...] = (1, 2, 4, 8)
    embed_ch: int = 512
    attn_res: Tuple[int, ...] = (16,)
    num_res_blocks: int = 2
    T: int = 1000                       # training schedule length (config.py `T`)
    beta_scheduler: str = "linear"
    gen_type: str = "ddpm"              # 'ddpm' | 'ddim'
    fp16: bool = True                   # reference autocast flag (inert on CPU)
    batch_size: int = 1
    # operand type of the convs / Linears / attention: "f32" (exact-fp32 MFMA), "bf16" (BASELINE config 4) or "f16"
    # (IEEE half: the reference's own GPU arithmetic under autocast, config_parm.py:40); 16-bit modes accumulate in
    # fp32.  Not part of the checkpoint name.
    compute_dtype: str = field(default="f32", compare=False)
    name: str = field(default="", compare=False)

    def __post_init__(self):
        if self.patch_size not in (32, 64, 128):
            raise NotImplementedError("Patch size not in [32, 64, 128]")
        if self.rna_slc not in DOWN_Z_KERNEL:
            raise ValueError(f"rna_slc {self.rna_slc} not in {sorted(DOWN_Z_KERNEL)}")
        if self.compute_dtype not in ("f32", "bf16", "f16"):
            raise ValueError(f"compute_dtype {self.compute_dtype!r}")
        if self.stain not in ("DAPI", "PolyT", "all"):
            raise ValueError(f"stain {self.stain!r}")
        if not self.name:
            self.name = (f"{self.mouse}_{self.patch_size}_{self.rna_num}_"
                         f"{self.stain}_{self.rna_slc}_{self.method}")

    # ---- derived quantities (reference: unet_ours.py:103-104, config.py:293-294,308) ----
    @property
    def z_size(self) -> int:
        return math.ceil(self.rna_slc / 2)

    @property
    def n_stain(self) -> int:
        return 2 if self.stain == "all" else 1

    @property
    def in_channels(self) -> int:
        return self.z_size * self.n_stain

    @property
    def gn_sz(self) -> int:                 # config_parm.py:47
        return self.patch_size // 16

    @property
    def gene_hidden(self) -> int:           # unet_ours.py:282 -- gn_sz^2 * rna_slc
        return self.gn_sz * self.gn_sz * self.rna_slc

    @property
    def rna_widths(self) -> Tuple[int, ...]:
        return (self.rna_num,) + RNA_PYRAMID_TAIL

    @property
    def down_z_kernel(self) -> int:
        return DOWN_Z_KERNEL[self.rna_slc]


def prep_config_parm(pth, bat, size, gpus, stain, mouse, nrna, srna=4, method="ours",
                     is_test=False) -> PathConfig:
    """Same positional signature as the reference's config_parm.prep_config_parm."""
    return PathConfig(patch_size=size, rna_slc=srna, stain=stain, rna_num=nrna, mouse=mouse,
                      method=method, batch_size=bat)


def parse_ckpt_dir_name(name: str) -> PathConfig:
    """'{mouse}_{size}_{nrna}_{stain}_{srna}_{method}'  (reference: test_brn.py:337-338)."""
    parts = name.split("_")
    if len(parts) < 6:
        raise ValueError(f"cannot parse checkpoint directory name {name!r}")
    mouse, size, nrna, stain, srna = parts[0], int(parts[1]), int(parts[2]), parts[3], int(parts[4])
    method = "_".join(parts[5:])
    return PathConfig(patch_size=size, rna_slc=srna, stain=stain, rna_num=nrna, mouse=mouse,
                      method=method)
