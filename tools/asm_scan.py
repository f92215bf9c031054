#!/usr/bin/env python3
"""Compiler-hygiene scan of the HIP kernels (no GPU needed): compiles every csrc/*.hip to gfx950 assembly and flags, per kernel,
the patterns that cost this repository the most before they were found (profiles/r03_conv1_glue_notes.txt):
  * v_div_scale_f32  -- IEEE division sequences (`__frcp_rn`, `x / y`, libm): ~10 VALU instructions each; fine in an fp32 parity
                        kernel, a VALU bound in a 16-bit streaming pass;
  * v_readlane / v_writelane -- SGPR spills to VGPR lanes (too many wave-uniform values live: unrolled weight loads);
  * scratch_*        -- VGPR spills (a gated epilogue that spilled ran 4 x slower);
  * v_mov share      -- register copies from conditionally defined arrays (loads under a wave-uniform `if`);
plus the VGPR count (waves per SIMD = 512 / VGPRs).
   python3 tools/asm_scan.py [--all]        (default: only kernels that trip a threshold)"""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tera-mind_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=fast", "--cuda-device-only", "-S"]


def main():
    show_all = "--all" in sys.argv
    srcs = [("tm_kernels.hip", []), ("tm_conv_bf16.hip", []), ("tm_conv_bf16.hip", ["-DTM_H16_F16"]), ("tm_attn.hip", []),
            ("tm_sampler.hip", ["-ffp-contract=off"]), ("tm_io.hip", []), ("tm_train.hip", [])]
    print(f"{'source':22s} {'kernel':84s} {'instr':>6s} {'VGPR':>5s} {'div':>4s} {'lane':>5s} {'scr':>4s} {'v_mov':>6s}")
    with tempfile.TemporaryDirectory() as tmp:
        for src, extra in srcs:
            out = os.path.join(tmp, src + "".join(extra).replace("-", "_") + ".s")
            r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-o", out, os.path.join(CSRC, src)], capture_output=True, text=True)
            if r.returncode != 0:
                print(f"{src}: compile failed\n{r.stderr[-2000:]}")
                continue
            s = open(out).read()
            for m in re.finditer(r"^(_Z\w+):[^\n]*\n", s, re.M):
                name = m.group(1)
                body = s[m.end():s.find(".Lfunc_end", m.end())]
                c = Counter(re.findall(r"^\s+([a-z]\w+)", body, re.M))
                tot = sum(c.values())
                div = c.get("v_div_scale_f32", 0)
                lane = c.get("v_readlane_b32", 0) + c.get("v_writelane_b32", 0)
                scr = sum(v for k, v in c.items() if k.startswith("scratch_"))
                mov = c.get("v_mov_b32_e32", 0) + c.get("v_mov_b64_e32", 0)
                vg = re.search(re.escape(name) + r"\.num_vgpr, (\d+)", s)
                if not vg:
                    continue                                  # a device function, not a kernel
                if show_all or div > 8 or lane > 40 or scr > 0 or (tot > 800 and mov > 0.15 * tot):
                    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                    dn = re.sub(r"\(.*\)$", "", dn).replace("tmk::", "").replace("void ", "")
                    print(f"{(src + ' ' + ' '.join(extra))[:22]:22s} {dn[:84]:84s} {tot:6d} {vg.group(1):>5s} {div:4d} {lane:5d} {scr:4d} {mov:6d}")


if __name__ == "__main__":
    main()
