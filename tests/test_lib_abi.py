"""CPU: the C-ABI library loads (no GPU needed for dlopen) and exports every function that
include/teramind_hip.h declares; argument validation paths that never touch the device."""
import ctypes as C
import os
import re

import pytest

import util
from teramind_amd import _lib


def header_functions():
    src = open(os.path.join(util.ROOT, "include", "teramind_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tm_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    names = header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"{n} declared in teramind_hip.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), "ctypes signature table out of sync with the header"
    assert L.tm_version() == 1


def test_error_paths_without_device():
    L = _lib.lib()
    cfg = _lib.TmConfig()
    cfg.patch_size, cfg.rna_slc, cfg.n_stain, cfg.rna_num, cfg.net_ch = 48, 4, 2, 229, 64
    cfg.embed_ch, cfg.attn_res, cfg.num_res_blocks = 512, 16, 2
    for i, v in enumerate((1, 2, 4, 8)):
        cfg.ch_mult[i] = v
    h = C.c_void_p(0)
    assert L.tm_model_create(C.byref(cfg), C.byref(h)) == -1           # TM_ERR_ARG: unsupported patch size
    assert b"patch_size" in L.tm_last_error()
    cfg.patch_size = 64
    assert L.tm_model_create(C.byref(cfg), C.byref(h)) == 0
    assert L.tm_model_num_params(h) == 399
    assert L.tm_model_param_key(h, 0) == b"time_embed.time_embed.0.weight"
    shp = (C.c_int64 * 2)(3, 3)
    buf = (C.c_float * 9)()
    assert L.tm_model_load_param(h, b"no.such.key", buf, shp, 2, 0) == -3      # TM_ERR_KEY
    assert L.tm_model_load_param(h, b"time_embed.time_embed.0.weight", buf, shp, 2, 0) == -3   # shape mismatch
    assert L.tm_model_finalize(h) == -3                                          # strict: missing keys
    assert b"missing key" in L.tm_last_error()
    assert L.tm_workspace_bytes(h, 1, 2, 2, 0) == 0                              # not finalized
    assert L.tm_model_destroy(h) == 0


def test_product_path_has_no_cpu_fallback():
    import torch
    from teramind_amd.config import PathConfig
    from teramind_amd.unet import BeatGANsUNetModel
    with pytest.raises(RuntimeError):
        BeatGANsUNetModel(PathConfig(), "cpu")
    # nothing under the package imports the oracle
    pkg = os.path.join(util.ROOT, "tera-mind_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            assert "oracle" not in open(os.path.join(pkg, f)).read().replace("CPU oracle", ""), f
