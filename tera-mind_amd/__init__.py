"""MI355X-native denoising hot path of Tera-MIND.

Python host side: mirrors the reference's call surface for this path and calls the C-ABI
HIP library `csrc/libteramind_hip.so` (include/teramind_hip.h).  There is no CPU fallback:
anything that computes raises if the library is missing.
"""
from .config import PathConfig, prep_config_parm, parse_ckpt_dir_name  # noqa: F401

__version__ = "0.1.0"
