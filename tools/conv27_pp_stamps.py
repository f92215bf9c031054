#!/usr/bin/env python3
"""Diagnostic for the ping-pong conv27 kernel: per wave group, where do the cycles of a segment go?  Runs single launches
(tm_op_conv27_time) on the diagnostic library (csrc `make diag`) and prints, per layer and per group, the median over
workgroups of the cycles per segment spent in: MFMA issue | wait at the barrier behind the MFMAs | LDS-DMA issue |
ds_reads until landed | wait at the barrier in front of the MFMAs, plus prologue / loop / epilogue cycles.
The stamps go to a buffer of their own; the product library contains no stamp code; the diagnostic build's run time is
not a measurement (its stamps forbid overlaps the real kernel has): read SHARES."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["TM_LIB_PATH"] = os.path.join(ROOT, "tera-mind_amd", "csrc", "libteramind_hip_diag.so")

LAYERS = [("enc2 c2 256>256 S16", 256, 256, 16, 625, 1, 0, 0), ("enc3 c2 512>512 S8", 512, 512, 8, 625, 1, 0, 0),
          ("dec1 c1 448>128 S32", 448, 128, 32, 400, 0, 1, 0), ("dec0 c1 160>64 S64", 160, 64, 64, 400, 0, 1, 0),
          ("up3 c1 512>512 S8 ups", 512, 512, 8, 400, 0, 0, 1)]


def main():
    import numpy as np
    import torch
    import teramind_amd  # noqa: F401
    from teramind_amd import _lib
    L = _lib.lib()
    L.tm_diag_stamps.restype, L.tm_diag_stamps.argtypes = C.c_int, [C.c_void_p, C.c_uint]
    L.tm_diag_stamp_count.restype, L.tm_diag_stamp_count.argtypes = C.c_int, []
    dev = torch.device("cuda:0")
    torch.zeros(1, device=dev)
    st = _lib.current_stream_ptr()
    ms = C.c_float(0)
    cap = 200_000
    buf = torch.zeros((cap, 16), dtype=torch.int64, device=dev)
    names = ["mfma", "wait_B", "-", "load", "wait_A"]
    for (name, cin, cout, S, N, res, fused, ups) in LAYERS:
        _lib.check(L.tm_op_conv27_time(N, cin, cout, S, 1, 8, ups, res, fused, 1, C.byref(ms), st), "warm")
        buf.zero_()
        assert L.tm_diag_stamps(C.c_void_p(buf.data_ptr()), cap) == 0
        _lib.check(L.tm_op_conv27_time(N, cin, cout, S, 1, 8, ups, res, fused, 1, C.byref(ms), st), "timed")
        torch.cuda.synchronize()
        n = L.tm_diag_stamp_count()
        L.tm_diag_stamps(None, 0)
        a = buf[:min(n, cap)].cpu().numpy()
        cbp = int(a[0, 6]) // 1000000
        nseg = (cbp * (1 if ups else 2)) * (4 if ups else 3)
        print(f"== {name}: {len(a)} slots, {cbp} pairs, {nseg} segments per workgroup, {ms.value:.3f} ms per launch (diagnostic build)")
        for g in (0, 1):
            s = a[((a[:, 6] % 10) // 4) == g]
            if len(s) == 0:
                continue
            pro, loop, epi = s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2]
            parts = [np.median(s[:, 8 + i]) / nseg for i in range(5)]
            print(f"   group {g}: prologue {np.median(pro):7.0f}  loop {np.median(loop):9.0f} ({np.median(loop) / nseg:6.0f} per segment pair of intervals)"
                  f"  epilogue {np.median(epi):7.0f} | per segment: " + "  ".join(f"{nm} {v:6.0f}" for nm, v in zip(names, parts)))


if __name__ == "__main__":
    main()
