"""ctypes binding of the C-ABI HIP library (include/teramind_hip.h).

There is deliberately no fallback: `lib()` raises if `csrc/libteramind_hip.so` is missing or
fails to load, and every compute entry point raises `RuntimeError` with the library's
`tm_last_error()` text on a non-zero return code.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# TM_LIB_PATH: diagnostic builds only (csrc `make diag`: the same library with in-kernel time stamps, tools/conv27_stamps.py)
LIB_PATH = os.environ.get("TM_LIB_PATH") or os.path.join(_HERE, "csrc", "libteramind_hip.so")

c_void_p, c_int, c_size_t, c_char_p, c_float = C.c_void_p, C.c_int, C.c_size_t, C.c_char_p, C.c_float
c_i64_p = C.POINTER(C.c_int64)


class TmConfig(C.Structure):
    _fields_ = [("patch_size", C.c_int32), ("rna_slc", C.c_int32), ("n_stain", C.c_int32),
                ("rna_num", C.c_int32), ("net_ch", C.c_int32), ("ch_mult", C.c_int32 * 4),
                ("embed_ch", C.c_int32), ("attn_res", C.c_int32), ("num_res_blocks", C.c_int32),
                ("vis_only", C.c_int32), ("dtype", C.c_int32)]


class TmStepCoefs(C.Structure):
    _fields_ = [("sqrt_recip_alphas_cumprod", c_float), ("sqrt_recipm1_alphas_cumprod", c_float),
                ("posterior_mean_coef1", c_float), ("posterior_mean_coef2", c_float),
                ("sigma", c_float), ("sqrt_alpha_bar_prev", c_float),
                ("sqrt_one_minus_alpha_bar_prev", c_float)]


class TmProfStats(C.Structure):
    _fields_ = [("launches", C.c_uint64), ("total_ms", C.c_double), ("nominal_flops", C.c_double),
                ("executed_flops", C.c_double), ("alg_bytes", C.c_double)]


# name -> (restype, argtypes); mirrors include/teramind_hip.h one to one
SIGNATURES = {
    "tm_version": (c_int, []),
    "tm_last_error": (c_char_p, []),
    "tm_model_create": (c_int, [C.POINTER(TmConfig), C.POINTER(c_void_p)]),
    "tm_model_load_param": (c_int, [c_void_p, c_char_p, c_void_p, c_i64_p, c_int, c_int]),
    "tm_model_finalize": (c_int, [c_void_p]),
    "tm_model_num_params": (c_int, [c_void_p]),
    "tm_model_param_key": (c_char_p, [c_void_p, c_int]),
    "tm_model_arena_bytes": (c_size_t, [c_void_p]),
    "tm_model_arena_ptr": (c_void_p, [c_void_p]),
    "tm_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int, c_int, c_int]),
    "tm_unet_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "tm_rna_level0_bytes": (c_size_t, [c_void_p, c_int, c_int, c_int]),
    "tm_rna_level0": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p]),
    "tm_unet_forward_level0": (c_int, [c_void_p] * 4 + [c_size_t] + [c_int] * 3 + [c_void_p] * 3 + [c_size_t, c_void_p]),
    "tm_rna_pyramid_bytes": (c_size_t, [c_void_p, c_int, c_int, c_int]),
    "tm_rna_pyramid": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "tm_unet_forward_rna": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_int,
                                    c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "tm_sampler_step": (c_int, [C.POINTER(TmStepCoefs), c_void_p, c_void_p, c_void_p, c_void_p,
                                c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "tm_pad_patchify": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "tm_gene_attn_workspace_bytes": (c_size_t, [c_void_p, c_int]),
    "tm_gene_attn": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "tm_model_destroy": (c_int, [c_void_p]),
    "tm_gene_tile_dense": (c_int, [c_void_p, c_void_p, C.c_int64] + [c_int] * 6 + [c_void_p, c_void_p]),
    "tm_blosc_decompress": (c_int, [c_void_p, c_size_t, c_void_p, c_size_t, C.POINTER(c_size_t)]),
    "tm_profile_enable": (c_int, [c_void_p, c_int]),
    "tm_profile_collect": (c_int, [c_void_p, C.POINTER(TmProfStats)]),
    "tm_op_to_cb8": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "tm_op_from_cb8": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "tm_op_conv_mfma": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 9 + [c_void_p]),
    "tm_op_conv27_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p] * 2 + [c_int] * 2 + [c_void_p]),
    "tm_op_conv27_fused": (c_int, [c_void_p] * 7 + [c_int] * 7 + [c_void_p]),
    "tm_op_conv27_time": (c_int, [c_int] * 10 + [c_void_p, c_void_p]),
    "tm_op_conv1_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 8 + [c_void_p] * 4),
    "tm_op_conv1_concat": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p] + [c_int] * 8 + [c_void_p]),
    "tm_op_prep_h16": (c_int, [c_void_p] * 3 + [c_int] * 7 + [c_void_p, c_int, c_int, c_void_p, c_void_p, C.c_long] +
                       [c_int] * 4 + [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "tm_op_prep_train": (c_int, [c_void_p] * 5 + [c_float, c_int, c_void_p] + [c_int] * 4 + [c_void_p]),
    "tm_op_prep_bwd": (c_int, [c_void_p] * 6 + [c_float, c_int] + [c_void_p] * 4 + [c_int] * 4 + [c_void_p]),
    "tm_op_conv_dgrad": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "tm_op_conv_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "tm_op_ew": (c_int, [c_int] + [c_void_p] * 5 + [C.c_long, c_void_p]),
    "tm_op_gemm_f32": (c_int, [c_void_p] * 4 + [c_int] * 3 + [c_void_p] + [c_int] * 3 + [c_float, c_void_p]),
    "tm_op_rows": (c_int, [c_int] + [c_void_p] * 5 + [C.c_long, c_int, c_void_p]),
    "tm_op_resample": (c_int, [c_void_p, c_void_p] + [c_int] * 5 + [c_void_p]),
    "tm_op_sumsq": (c_int, [c_void_p, C.c_long, c_void_p, c_void_p]),
    "tm_op_adam": (c_int, [c_void_p] * 4 + [C.c_long] + [c_float] * 5 + [c_int, c_float, c_void_p]),
    "tm_op_modnorm": (c_int, [c_void_p] * 5 + [c_int] * 4 + [c_void_p]),
    "tm_op_modnorm_bwd": (c_int, [c_void_p] * 8 + [c_int] * 4 + [c_void_p]),
    "tm_op_window_attn_train": (c_int, [c_void_p] * 12 + [c_int] * 4 + [c_void_p]),
    "tm_op_window_attn": (c_int, [c_void_p] * 6 + [c_int] * 5 + [c_void_p]),
    "tm_op_conv_direct": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 13 + [c_void_p]),
}

_lib = None


def lib():
    """The loaded library (loads on first use).  Raises if it is missing: no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  teramind_amd has no CPU fallback.")
        # torch first: it ships its own libamdhip64, and the library must bind to THAT runtime (the one that owns the
        # tensors' device context) -- loaded the other way round the process ends up with two HIP runtimes and the
        # library's hipMalloc reports "no ROCm-capable device"
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)       # AttributeError here = header/library mismatch
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().tm_last_error().decode(errors="replace")
        raise RuntimeError(f"libteramind_hip {what} failed (code {rc}): {msg}")


def current_stream_ptr():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device / host pointer of a contiguous torch tensor (or None)."""
    if t is None:
        return c_void_p(0)
    assert t.is_contiguous()
    return c_void_p(t.data_ptr())
