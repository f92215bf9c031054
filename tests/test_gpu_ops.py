"""-m gpu: single-kernel parity through the C-ABI (tm_op_*), HIP vs torch CPU fp32.
Integer-valued operands make the fp32 result exact, so those cases are compared bit for bit
(catches any indexing / fragment-layout error without a tolerance to hide behind)."""
import pytest
import torch
import torch.nn.functional as F

import util
from teramind_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_cb8_roundtrip():
    x = util.rand_int((3, 13, 2, 8, 8), -5, 5, 0).to(DEV)
    y = util.to_cb8(x)
    assert y.shape == (3, 2, 2, 8, 8, 8)
    assert torch.equal(y[:, 1, :, :, :, 5:], torch.zeros_like(y[:, 1, :, :, :, 5:]))      # zero padded slots
    assert torch.equal(util.from_cb8(y, 13), x)


# (N, Cin, Cout, S)
CONV3_CASES = [(2, 16, 64, 8), (3, 24, 128, 8), (5, 13, 32, 8), (1, 8, 64, 16), (2, 40, 192, 16),
               (1, 72, 64, 32), (1, 8, 64, 64), (2, 16, 128, 64)]


@pytest.mark.parametrize("N,Cin,Cout,S", CONV3_CASES)
@pytest.mark.parametrize("variant", [1, 2])
def test_conv3_mfma_exact_integers(N, Cin, Cout, S, variant):
    x = util.rand_int((N, Cin, 2, S, S), -3, 3, 1)
    w = util.rand_int((Cout, Cin, 3, 3, 3), -2, 2, 2)
    b = util.rand_int((Cout,), -4, 4, 3)
    ref = F.conv3d(x, w, b, padding=1)
    got, raw = util.conv_mfma(x.to(DEV), w, b, 3, variant)
    assert torch.equal(got.cpu(), ref), util.report("conv3", got, ref)
    if Cout % 8:
        assert float(raw[:, -1, ..., Cout % 8:].abs().max()) == 0.0


@pytest.mark.parametrize("N,Cin,Cout,S", [(2, 96, 64, 16), (1, 741, 512, 8), (1, 224, 64, 64)])
def test_conv3_mfma_random(N, Cin, Cout, S):
    g = torch.Generator().manual_seed(7)
    x = torch.randn((N, Cin, 2, S, S), generator=g)
    w = torch.randn((Cout, Cin, 3, 3, 3), generator=g) / (Cin * 27) ** 0.5
    b = torch.randn((Cout,), generator=g)
    ref = F.conv3d(x, w, b, padding=1)
    got, _ = util.conv_mfma(x.to(DEV), w, b, 3)
    # fp32 accumulation-order noise only: K <= 20007 products of O(1)*O(K^-1/2)
    assert torch.allclose(got.cpu(), ref, atol=2e-5, rtol=1e-5), util.report("conv3 random", got, ref)


@pytest.mark.parametrize("N,Cin,Cout,Z,S", [(2, 229, 128, 2, 8), (1, 64, 1792, 2, 16), (3, 13, 32, 2, 8),
                                            (1, 96, 64, 2, 64), (5, 512, 2048, 2, 8)])
@pytest.mark.parametrize("variant", [1, 2])
def test_conv1_mfma_exact_integers(N, Cin, Cout, Z, S, variant):
    x = util.rand_int((N, Cin, Z, S, S), -3, 3, 4)
    w = util.rand_int((Cout, Cin, 1, 1, 1), -2, 2, 5)
    b = util.rand_int((Cout,), -4, 4, 6)
    ref = F.conv3d(x, w, b)
    got, _ = util.conv_mfma(x.to(DEV), w, b, 1, variant)
    assert torch.equal(got.cpu(), ref), util.report("conv1", got, ref)


DIRECT_CASES = [
    # N, Cin, Cout, Zin, S, k, pad, silu, up2
    (3, 2, 64, 2, 64, (1, 3, 3), (0, 1, 1), False, False),     # stem
    (2, 64, 2, 2, 64, (1, 3, 3), (0, 1, 1), False, False),     # head
    (3, 229, 229, 4, 4, (3, 3, 3), (0, 1, 1), False, True),    # down_z + upsample
    (2, 229, 128, 2, 8, (1, 3, 3), (0, 1, 1), True, True),     # pyramid level
]


@pytest.mark.parametrize("N,Cin,Cout,Zin,S,k,pad,silu,up2", DIRECT_CASES)
def test_conv_direct(N, Cin, Cout, Zin, S, k, pad, silu, up2):
    g = torch.Generator().manual_seed(11)
    x = torch.randn((N, Cin, Zin, S, S), generator=g)
    w = torch.randn((Cout, Cin) + k, generator=g) / (Cin * k[0] * k[1] * k[2]) ** 0.5
    b = torch.randn((Cout,), generator=g)
    xi = x * torch.sigmoid(x) if silu else x
    ref = F.conv3d(xi, w, b, padding=pad)
    if up2:
        ref = ref.repeat_interleave(2, -2).repeat_interleave(2, -1)
    got = util.conv_direct(x.to(DEV), w, b, pad, silu, up2)
    assert torch.allclose(got.cpu(), ref, atol=2e-5, rtol=1e-5), util.report("direct", got, ref)


@pytest.mark.parametrize("N,Cin,Cout,S", [(3, 229, 128, 8), (2, 128, 64, 16), (2, 64, 32, 32), (5, 13, 40, 8)])
@pytest.mark.parametrize("up2", [False, True])
def test_conv_inplane_mfma_exact_integers(N, Cin, Cout, S, up2):
    """1x3x3 in-plane conv (RNA pyramid) + fused nearest-x2 store."""
    x = util.rand_int((N, Cin, 2, S, S), -3, 3, 21)
    w = util.rand_int((Cout, Cin, 1, 3, 3), -2, 2, 22)
    b = util.rand_int((Cout,), -4, 4, 23)
    ref = F.conv3d(x, w, b, padding=(0, 1, 1))
    if up2:
        ref = ref.repeat_interleave(2, -2).repeat_interleave(2, -1)
    got, _ = util.conv_mfma(x.to(DEV), w, b, 3, zmode=1, up2=up2)
    assert torch.equal(got.cpu(), ref), util.report("conv 1x3x3", got, ref)


@pytest.mark.parametrize("N,Cin,Cout", [(3, 229, 229), (9, 16, 64), (17, 24, 72)])
def test_conv_validz_mfma_exact_integers(N, Cin, Cout):
    """3x3x3 valid-in-z conv on 4x4x4 gene volumes (down_z) + fused nearest-x2 store."""
    x = util.rand_int((N, Cin, 4, 4, 4), -3, 3, 31)
    w = util.rand_int((Cout, Cin, 3, 3, 3), -2, 2, 32)
    b = util.rand_int((Cout,), -4, 4, 33)
    ref = F.conv3d(x, w, b, padding=(0, 1, 1)).repeat_interleave(2, -2).repeat_interleave(2, -1)
    got, raw = util.conv_mfma(x.to(DEV), w, b, 3, zmode=2, up2=True)
    assert got.shape == (N, Cout, 2, 8, 8)
    assert torch.equal(got.cpu(), ref), util.report("down_z", got, ref)
    if Cout % 8:
        assert float(raw[:, -1, ..., Cout % 8:].abs().max()) == 0.0


BF16_CASES = [(2, 16, 64, 8), (17, 24, 128, 8), (3, 13, 40, 8), (9, 229, 512, 8), (5, 40, 192, 16), (3, 96, 64, 16),
              (1, 72, 64, 32), (2, 16, 128, 32), (1, 8, 64, 64), (1, 24, 128, 64)]


@pytest.mark.parametrize("N,Cin,Cout,S", BF16_CASES)
@pytest.mark.parametrize("waves", [4, 8])
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_conv27_bf16_exact_integers(N, Cin, Cout, S, waves, dtype):
    """Small integers are exact in bf16 / fp16 and their products/sums exact in fp32: bit-exact check of the
    16-bit MFMA kernel's fragment layout, pair padding, slot swizzle and tiling -- in BOTH workgroup forms (the launcher
    picks 4 waves for grids that would leave CUs idle and 8 waves otherwise; every case here is forced through each)
    and both element types."""
    x = util.rand_int((N, Cin, 2, S, S), -3, 3, 41)
    w = util.rand_int((Cout, Cin, 3, 3, 3), -2, 2, 42)
    b = util.rand_int((Cout,), -4, 4, 43)
    ref = F.conv3d(x, w, b, padding=1)
    got, raw = util.conv27_bf16(x.to(DEV), w, b, dtype, waves)
    assert torch.equal(got.cpu(), ref), util.report("conv27 " + dtype, got, ref)
    if Cout % 8:
        assert float(raw[:, -1, ..., Cout % 8:].abs().max()) == 0.0


@pytest.mark.parametrize("N,Cin,Cout,S", [(5, 24, 64, 8), (3, 40, 192, 16), (2, 16, 128, 32), (1, 24, 128, 64), (9, 229, 512, 8)])
@pytest.mark.parametrize("waves", [4, 8])
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_conv27_16bit_stream_epilogue(N, Cin, Cout, S, waves, dtype):
    """The model's form of the second ResBlock conv in the 16-bit modes: 16-bit CB8 residual in, 16-bit CB8 result out
    (16-byte accesses after a v_permlane32_swap of the accumulator quads).  Integer operands: the fp32 value
    conv + bias + res is exact, so the only rounding is the final RNE to the 16-bit type -- bit-exact against torch."""
    td = util.H16[dtype][1]
    x = util.rand_int((N, Cin, 2, S, S), -3, 3, 71)
    w = util.rand_int((Cout, Cin, 3, 3, 3), -2, 2, 72)
    b = util.rand_int((Cout,), -4, 4, 73)
    res = util.rand_int((N, Cout, 2, S, S), -100, 100, 74)
    ref = (F.conv3d(x, w, b, padding=1) + res).to(td).float()
    got, raw = util.conv27_bf16(x.to(DEV), w, b, dtype, waves, res=res.to(DEV), out16=True)
    assert torch.equal(got.cpu(), ref), util.report("conv27 stream " + dtype, got, ref)


@pytest.mark.parametrize("N,Cin,Cout,S", [(328, 16, 512, 8), (135, 16, 256, 16), (70, 24, 128, 32), (289, 16, 64, 16)])
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_conv27_16bit_tail_split_launch(N, Cin, Cout, S, dtype):
    """Launches whose last round of 8-wave workgroups would occupy at most half the CUs are split: full rounds on the
    ping-pong 8-wave kernel, the remaining tiles as 4-wave workgroups of half the voxels (launch_conv27_bf16, TAIL SPLIT).
    Shapes here produce 256 + a short tail of tiles in every tile geometry (patch groups of 8 / 2 / 1, two tile rows, an odd
    patch count whose last 8-wave group is half empty): the union must be exactly the layer -- bit-exact on integers, with
    the 16-bit residual / output epilogue."""
    td = util.H16[dtype][1]
    x = util.rand_int((N, Cin, 2, S, S), -3, 3, 91)
    w = util.rand_int((Cout, Cin, 3, 3, 3), -2, 2, 92)
    b = util.rand_int((Cout,), -4, 4, 93)
    res = util.rand_int((N, Cout, 2, S, S), -100, 100, 94)
    ref = (F.conv3d(x, w, b, padding=1) + res).to(td).float()
    got, _ = util.conv27_bf16(x.to(DEV), w, b, dtype, 0, res=res.to(DEV), out16=True)
    assert torch.equal(got.cpu(), ref), util.report("conv27 tail split " + dtype, got, ref)


@pytest.mark.parametrize("N,Cin,Cout,Z,S", [(2, 229, 1792, 2, 8), (3, 13, 40, 2, 8), (1, 96, 64, 2, 64), (7, 128, 64, 2, 16)])
@pytest.mark.parametrize("waves", [4, 8])
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_conv1_16bit_stream_epilogue(N, Cin, Cout, Z, S, waves, dtype):
    """x <- x + gate * Linear(.) with 16-bit gate, residual and output (AttnBlock, model/MBAblocks.py:486-489)."""
    td = util.H16[dtype][1]
    x = util.rand_int((N, Cin, Z, S, S), -3, 3, 81)
    w = util.rand_int((Cout, Cin, 1, 1, 1), -2, 2, 82)
    b = util.rand_int((Cout,), -4, 4, 83)
    res = util.rand_int((N, Cout, Z, S, S), -100, 100, 84)
    gate = util.rand_int((N, Cout, Z, S, S), -2, 2, 85)
    ref = (res + gate * F.conv3d(x, w, b)).to(td).float()
    got, _ = util.conv1_bf16(x.to(DEV), w, b, False, dtype, waves, res=res.to(DEV), gate=gate.to(DEV), out16=True)
    assert torch.equal(got.cpu(), ref), util.report("conv1 stream " + dtype, got, ref)


@pytest.mark.parametrize("N,Cin,Cout,Z,S", [(2, 229, 1792, 2, 8), (3, 13, 40, 2, 8), (1, 96, 64, 2, 64), (5, 256, 256, 2, 16)])
@pytest.mark.parametrize("waves", [4, 8])
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_conv1_16bit_stream_epilogue_plain_and_gelu(N, Cin, Cout, Z, S, waves, dtype):
    """The stream epilogue kinds without side inputs (conv1_epilogue_stream<1> / <2>: q, k / v, adaLN and fc1 of the AttnBlock):
    16-bit output of the plain Linear bit-exact on integers (ragged cout blocks, ragged voxel tiles); with tanh-GELU (hardware
    exp2 / rcp, then rounded to the 16-bit type) to one 16-bit ulp of torch's tanh-GELU of the exact integer pre-activation."""
    td = util.H16[dtype][1]
    x = util.rand_int((N, Cin, Z, S, S), -3, 3, 61)
    w = util.rand_int((Cout, Cin, 1, 1, 1), -2, 2, 62)
    b = util.rand_int((Cout,), -4, 4, 63)
    pre = F.conv3d(x, w, b)
    got, _ = util.conv1_bf16(x.to(DEV), w, b, False, dtype, waves, out16=True)
    assert torch.equal(got.cpu(), pre.to(td).float()), util.report("conv1 stream plain " + dtype, got, pre)
    # GELU on small pre-activations (|v| up to a few units: where the curve bends)
    ws = w * 0.0
    ws[:, :1] = w[:, :1].sign()                                # one input channel: pre-activation = +-x0 + b in [-7, 7]
    pre = F.conv3d(x, ws, b)
    ref = F.gelu(pre, approximate="tanh")
    got, _ = util.conv1_bf16(x.to(DEV), ws, b, True, dtype, waves, out16=True)
    ulp = (2.0 ** -7 if dtype == "bf16" else 2.0 ** -10) * ref.abs().clamp_min(2.0 ** -14)
    assert ((got.cpu() - ref).abs() <= 1.01 * ulp + 1e-7).all(), util.report("conv1 stream gelu " + dtype, got, ref)


@pytest.mark.parametrize("b,p1,p2,cins,flags,Cout,S", [
    (2, 3, 3, (64, 32), (0, 0), 64, 16),              # encoder block: cat(h, rna), plain
    (1, 2, 4, (128, 64, 32), (1, 1, 1), 128, 8),      # first decoder block of a level: every source re-tiled
    (2, 3, 2, (40, 24, 229), (0, 1, 1), 512, 8),      # later decoder block: h plain, skip / rna re-tiled; odd block counts
    (3, 2, 2, (229,), (1,), 192, 8)])
@pytest.mark.parametrize("waves", [4, 8])
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_conv1_concat_collage_input_exact_integers(b, p1, p2, cins, flags, Cout, S, waves, dtype):
    """The skip conv reads th.cat((h, skip, rna), 1) -- and the collage re-tiling of the decoder's sources -- in place
    (model/unet_ours.py:325-341,384,418): exact-integer check against torch.cat + the oracle's to_collage."""
    import ctypes as C
    from oracle import teramind_cpu as tc
    Nd, Ne = b * (p1 - 1) * (p2 - 1), b * p1 * p2
    any_col = any(flags)
    N = Nd if any_col else Ne
    xs, parts = [], []
    for i, (c, f) in enumerate(zip(cins, flags)):
        x = util.rand_int((Ne if f else N, c, 2, S, S), -3, 3, 90 + i)
        xs.append(x)
        parts.append(tc.collage(x, b, p1, p2) if f else x)
    w = util.rand_int((Cout, sum(cins), 1, 1, 1), -2, 2, 95)
    bias = util.rand_int((Cout,), -4, 4, 96)
    ref = F.conv3d(torch.cat(parts, 1), w, bias)
    xc = [util.to_cb8(x.to(DEV)) for x in xs]
    ptrs = (C.c_void_p * len(xc))(*[t.data_ptr() for t in xc])
    cin = (C.c_int * len(xc))(*cins)
    col = (C.c_int * len(xc))(*flags)
    yc = torch.zeros((N, (Cout + 7) // 8, 2, S, S, 8), dtype=torch.float32, device=DEV)
    wh, bh = w.contiguous().float(), bias.contiguous().float()
    _lib.check(_lib.lib().tm_op_conv1_concat(ptrs, cin, col, len(xc), C.c_void_p(wh.data_ptr()), C.c_void_p(bh.data_ptr()),
                                             _lib.ptr(yc), N, Cout, 2, S, p1, p2, util.H16[dtype][0], waves,
                                             _lib.current_stream_ptr()), "tm_op_conv1_concat")
    got = util.from_cb8(yc, Cout)
    assert torch.equal(got.cpu(), ref), util.report("conv1 concat " + dtype, got, ref)


# (b, p1, p2, cins, collage flags, S, up2, mod, with norm)
PREP_CASES = [
    (2, 3, 3, (64, 32), (0, 0), 16, False, 0, True),            # encoder block input: cat(h, rna)
    (1, 2, 4, (128, 64, 32), (1, 1, 1), 8, False, 0, True),     # decoder block input, every source re-tiled
    (2, 3, 2, (40, 24, 229), (0, 1, 1), 8, False, 0, True),     # odd block counts, 3 pad channels inside the concat
    (5, 2, 2, (64,), (0,), 16, True, 0, True),                  # up block: nearest x2 of a single source
    (3, 2, 2, (256,), (0,), 8, False, 1, True),                 # out_layers of a Cout > 128 block: per-image scale / shift
    (2, 2, 2, (128,), (0,), 16, False, 2, True),                # AttnBlock modulate: per-voxel scale / shift, no SiLU
    (2, 2, 3, (229,), (1,), 8, False, 0, False),                # SiLU(cond) of the collage decoder: no norm
    (1, 2, 2, (512, 512, 229), (0, 1, 1), 8, False, 0, True),   # 157 channel blocks (the widest concat of the model)
    (3, 2, 2, (16,), (0,), 8, False, 0, True),                  # 2 channel blocks: two of the four waves own none (clamped loads)
    (2, 2, 2, (8, 8, 8), (0, 1, 0), 16, False, 2, True)]        # 3 blocks from three sources, per-voxel modulation


@pytest.mark.parametrize("b,p1,p2,cins,flags,S,up2,mod,norm", PREP_CASES)
@pytest.mark.parametrize("variant", [0, 1, 2, 3])
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_prep_h16_forms_vs_torch(b, p1, p2, cins, flags, S, up2, mod, norm, variant, dtype):
    """cat + to_collage / Upsample + LlamaRMSNorm + modulate + SiLU on the 16-bit stream (model/unet_ours.py:325-341,384,418,
    model/MBAblocks.py:21-43,254-261,356-367,484-489,608-614): every kernel form against fp32 torch on the same 16-bit
    inputs, to one 16-bit ulp (the forms differ in the order of the sum of squares and use the hardware exp / rcp)."""
    from oracle import teramind_cpu as tc
    if variant == 2 and sum((c + 7) // 8 for c in cins) > 32:
        pytest.skip("one-wave form holds at most 32 channel blocks")
    td = util.H16[dtype][1]
    g = torch.Generator().manual_seed(5)
    Ne, Nd = b * p1 * p2, b * (p1 - 1) * (p2 - 1)
    any_col = any(flags)
    N = Nd if any_col else Ne
    Ss = S // 2 if up2 else S
    xs, parts = [], []
    for c, f in zip(cins, flags):
        x = (torch.randn((Ne if f else N, c, 2, Ss, Ss), generator=g) * 1.5).to(td).float()
        xs.append(x)
        y = tc.collage(x, b, p1, p2) if f else x
        if up2:
            y = y.repeat_interleave(2, 3).repeat_interleave(2, 4)
        parts.append(y)
    xcat = torch.cat(parts, 1)
    Ct = xcat.shape[1]
    ws = [torch.randn((c,), generator=g) * 0.3 + 1.0 for c in cins] if norm else None
    ref = xcat
    if norm:
        ref = torch.cat(ws) [None, :, None, None, None] * (xcat * torch.rsqrt(xcat.pow(2).mean(1, keepdim=True) + 1e-6))
    per_image = N // b
    scale = shift = None
    if mod == 1:
        scale, shift = torch.randn((b, Ct), generator=g) * 0.5, torch.randn((b, Ct), generator=g) * 0.5
        ix = torch.arange(N) // per_image
        ref = ref * (1 + scale[ix][:, :, None, None, None]) + shift[ix][:, :, None, None, None]
    elif mod == 2:
        scale = (torch.randn(xcat.shape, generator=g) * 0.5).to(td).float()
        shift = (torch.randn(xcat.shape, generator=g) * 0.5).to(td).float()
        ref = ref * (1 + scale) + shift
    act = mod != 2
    if act:
        ref = F.silu(ref)
    got, raw, _ = util.prep_h16([x.to(DEV) for x in xs], cins, flags, b, p1, p2, S, up2=up2, norm_w=ws, mod=mod, scale=scale,
                                shift=shift, per_image=per_image, act=act, dtype=dtype, variant=variant, want_raw=True)
    ulp = 2.0 ** -7 if dtype == "bf16" else 2.0 ** -10
    o = 0
    for i, c in enumerate(cins):
        r = ref[:, o:o + c]
        err = (got[i].cpu() - r).abs()
        assert bool((err <= ulp * r.abs() + 1e-6).all()), util.report(f"prep {dtype} v{variant} src{i}", got[i], r)
        assert torch.equal(raw[i].cpu(), xcat[:, o:o + c]), "raw copy must be the gathered input, bit for bit"
        o += c


@pytest.mark.parametrize("cins,flags,mod", [((64, 32), (0, 0), 0), ((128, 64, 32), (1, 1, 1), 0), ((128,), (0,), 2), ((229,), (0,), 0)])
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_prep_h16_forms_bit_identical(cins, flags, mod, dtype):
    """The launcher picks the one-wave or the split form of prep_h16_kernel by the size of the call; both add the sum of
    squares in the same order, so the choice never changes a bit (a patch inside a tile-sized batch == the patch alone)."""
    td = util.H16[dtype][1]
    g = torch.Generator().manual_seed(9)
    b, p1, p2, S = 2, 3, 3, 16
    any_col = any(flags)
    Ne, N = b * p1 * p2, (b * (p1 - 1) * (p2 - 1) if any_col else b * p1 * p2)
    xs = [(torch.randn((Ne if f else N, c, 2, S, S), generator=g) * 2).to(td).float().to(DEV) for c, f in zip(cins, flags)]
    ws = [torch.randn((c,), generator=g) * 0.3 + 1.0 for c in cins]
    scale = shift = None
    if mod == 2:
        scale = (torch.randn((N, sum(cins), 2, S, S), generator=g) * 0.5).to(td).float()
        shift = (torch.randn((N, sum(cins), 2, S, S), generator=g) * 0.5).to(td).float()
    outs = [util.prep_h16(xs, cins, flags, b, p1, p2, S, norm_w=ws, mod=mod, scale=scale, shift=shift, per_image=N // b,
                          act=mod != 2, dtype=dtype, variant=v)[0] for v in (2, 3)]
    for a, c in zip(*outs):
        assert torch.equal(a, c)


@pytest.mark.parametrize("N,Cin,Cout,S", [(2, 16, 64, 8), (3, 24, 128, 4), (1, 72, 64, 16), (2, 8, 64, 32), (1, 128, 128, 32)])
@pytest.mark.parametrize("variant", [1, 2])
def test_conv3_upsampled_input_phase_form_exact_integers(N, Cin, Cout, S, variant):
    """Upsample (nearest x2) -> Conv3d(3, pad 1) of ResBlock(up=True) (model/MBAblocks.py:254-258, blocks.py:362-371)
    computed on the low-resolution tensor with per-phase 2 x 2 in-plane weights (sums of the 3 x 3 taps): integer operands
    make both associations exact, so the result must equal F.conv3d on the upsampled tensor bit for bit."""
    x = util.rand_int((N, Cin, 2, S, S), -3, 3, 61)
    w = util.rand_int((Cout, Cin, 3, 3, 3), -2, 2, 62)
    b = util.rand_int((Cout,), -4, 4, 63)
    ref = F.conv3d(x.repeat_interleave(2, 3).repeat_interleave(2, 4), w, b, padding=1)
    got, _ = util.conv_mfma(x.to(DEV), w, b, 3, variant=variant, zmode=3)
    assert torch.equal(got.cpu(), ref), util.report("conv3 upsampled-input", got, ref)


@pytest.mark.parametrize("N,Cin,Cout,S", [(2, 16, 128, 8), (3, 40, 128, 16), (1, 128, 128, 32), (2, 64, 256, 8), (1, 32, 256, 16)])
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_conv27_16bit_upsampled_input_phase_form_exact_integers(N, Cin, Cout, S, dtype):
    """The 16-bit twin of the upsampled-input conv: Upsample (nearest x2) -> Conv3d(3, pad 1) of ResBlock(up=True)
    (model/MBAblocks.py:254-258) on the low-resolution tensor with per-phase 2 x 2 weights.  Small integers: the summed
    weights (|w| <= 8) and every product are exact in bf16 / f16, so the result equals F.conv3d on the upsampled tensor."""
    x = util.rand_int((N, Cin, 2, S, S), -3, 3, 71)
    w = util.rand_int((Cout, Cin, 3, 3, 3), -2, 2, 72)
    b = util.rand_int((Cout,), -4, 4, 73)
    ref = F.conv3d(x.repeat_interleave(2, 3).repeat_interleave(2, 4), w, b, padding=1)
    got, _ = util.conv27_bf16(x.to(DEV), w, b, dtype, ups=True)
    assert torch.equal(got.cpu(), ref), util.report("conv27 16-bit upsampled-input " + dtype, got, ref)


@pytest.mark.parametrize("dtype,bound", [("bf16", 3.5e-3), ("f16", 4.5e-4)])
def test_conv27_16bit_upsampled_input_form_rounding_vs_tapwise_rounding(dtype, bound):
    """The upsampled-input form adds the 3 x 3 in-plane taps of each output phase into 2 x 2 weights in fp32 and rounds the SUM
    to the 16-bit type; the reference under autocast rounds each of the 27 taps and accumulates the products
    (round(w1 + w2) x  vs  round(w1) x + round(w2) x).  On random weights the two differ: this records the size.  Against
    F.conv3d in float64 on the upsampled tensor the two roundings are equally far from the exact-weight result (relative L2
    1.7e-3 bf16, 2.1e-4 f16 at these shapes) and 2.2e-3 / 2.8e-4 from each other; the bounds leave ~50 % headroom."""
    td = util.H16[dtype][1]
    g = torch.Generator().manual_seed(81)
    N, Cin, Cout, S = 2, 64, 128, 8
    x = torch.randn((N, Cin, 2, S, S), generator=g).to(td).float()
    w = torch.randn((Cout, Cin, 3, 3, 3), generator=g) / (Cin * 27) ** 0.5
    b = torch.zeros((Cout,))
    xu = x.double().repeat_interleave(2, 3).repeat_interleave(2, 4)
    exact = F.conv3d(xu, w.double(), b.double(), padding=1)
    tapwise = F.conv3d(xu, w.to(td).double(), b.double(), padding=1)
    got, _ = util.conv27_bf16(x.to(DEV), w, b, dtype, ups=True)
    rel = lambda a, r: float((a.double().cpu() - r).norm() / r.norm())
    e_ups, e_tap, e_between = rel(got, exact), rel(tapwise, exact), rel(got, tapwise)
    assert e_between < bound, (e_between, e_ups, e_tap)
    assert e_ups < 1.15 * e_tap, (e_ups, e_tap)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_conv27_16bit_residual_at_half_resolution(dtype):
    """res + conv with the residual stored at S/2 and read at (z, y >> 1, x >> 1): the residual of ResBlock(up=True) is the
    upsampled block input (model/MBAblocks.py:297)."""
    td = util.H16[dtype][1]
    N, Cin, Cout, S = 2, 24, 128, 16
    x = util.rand_int((N, Cin, 2, S, S), -3, 3, 75)
    w = util.rand_int((Cout, Cin, 3, 3, 3), -2, 2, 76)
    b = util.rand_int((Cout,), -4, 4, 77)
    res = util.rand_int((N, Cout, 2, S // 2, S // 2), -50, 50, 78)
    ref = (res.repeat_interleave(2, 3).repeat_interleave(2, 4) + F.conv3d(x, w, b, padding=1)).to(td).float()
    got, _ = util.conv27_bf16(x.to(DEV), w, b, dtype, res=res.to(DEV), out16=True, res_half=True)
    assert torch.equal(got.cpu(), ref), util.report("conv27 half-resolution residual " + dtype, got, ref)


@pytest.mark.parametrize("C_,S", [(64, 32), (128, 16), (256, 8), (40, 8)])
@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_prep_h16_downsample_form_vs_torch(C_, S, variant, dtype):
    """ResBlock(down=True) input (model/MBAblocks.py:254-258, blocks.py:389-403): LlamaRMSNorm -> SiLU at full resolution,
    then the 2 x 2 average of the activated tensor and (raw) of x itself; the 16-bit kernel and the generic one."""
    td = util.H16[dtype][1]
    g = torch.Generator().manual_seed(11)
    N = 3
    x = (torch.randn((N, C_, 2, 2 * S, 2 * S), generator=g) * 1.5).to(td).float()
    w = torch.randn((C_,), generator=g) * 0.3 + 1.0
    h = F.silu(w[None, :, None, None, None] * (x * torch.rsqrt(x.pow(2).mean(1, keepdim=True) + 1e-6)))
    pool = lambda t: F.avg_pool3d(t, (1, 2, 2))
    got, raw, _ = util.prep_h16([x.to(DEV)], (C_,), (0,), N, 2, 2, S, up2=2, norm_w=[w], act=True, dtype=dtype, variant=variant,
                                want_raw=True)
    ulp = 2.0 ** -7 if dtype == "bf16" else 2.0 ** -10
    for t, r in ((got[0].cpu(), pool(h)), (raw[0].cpu(), pool(x))):
        assert bool(((t - r).abs() <= ulp * r.abs() + 1e-6).all()), util.report(f"prep down {dtype} v{variant}", t, r)


def test_conv27_bf16_random_vs_bf16_rounded_reference():
    g = torch.Generator().manual_seed(17)
    N, Cin, Cout, S = 2, 741, 512, 8
    x = torch.randn((N, Cin, 2, S, S), generator=g)
    w = torch.randn((Cout, Cin, 3, 3, 3), generator=g) / (Cin * 27) ** 0.5
    b = torch.randn((Cout,), generator=g)
    ref = F.conv3d(x.bfloat16().float(), w.bfloat16().float(), b, padding=1)      # same rounded operands, fp32 math
    got, _ = util.conv27_bf16(x.to(DEV), w, b)
    assert torch.allclose(got.cpu(), ref, atol=3e-5, rtol=1e-5), util.report("conv27 bf16 random", got, ref)
    full = F.conv3d(x, w, b, padding=1)
    rel = ((got.cpu() - full).norm() / full.norm()).item()
    assert rel < 6e-3, rel                                                          # bf16 operand rounding: ~2^-9 relative


@pytest.mark.parametrize("N,Cin,Cout,Z,S", [(2, 229, 1792, 2, 8), (1, 64, 256, 2, 16), (3, 13, 40, 2, 8), (1, 96, 64, 2, 64),
                                            (5, 512, 2048, 2, 8), (2, 1253, 512, 2, 8), (7, 128, 64, 2, 16)])
@pytest.mark.parametrize("waves", [4, 8])
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_conv1_bf16_exact_integers(N, Cin, Cout, Z, S, waves, dtype):
    """16-bit Linear / 1x1 conv incl. ragged channel-pair stages and ragged voxel tiles, in both workgroup forms
    (two co-resident 4-wave workgroups per CU -- the default -- and one 8-wave workgroup) and both element types."""
    x = util.rand_int((N, Cin, Z, S, S), -3, 3, 51)
    w = util.rand_int((Cout, Cin, 1, 1, 1), -2, 2, 52)
    b = util.rand_int((Cout,), -4, 4, 53)
    ref = F.conv3d(x, w, b)
    got, _ = util.conv1_bf16(x.to(DEV), w, b, False, dtype, waves)
    assert torch.equal(got.cpu(), ref), util.report("conv1 " + dtype, got, ref)


@pytest.mark.parametrize("N,Cin,Cout,S,per_image", [(5, 24, 64, 8, 2), (3, 96, 128, 16, 1), (2, 40, 64, 32, 2), (1, 16, 128, 64, 1),
                                                    (17, 13, 128, 8, 4)])
@pytest.mark.parametrize("waves", [4, 8])
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_conv27_fused_norm_modulate_silu_epilogue(N, Cin, Cout, S, per_image, waves, dtype):
    """The ResBlock mid-section fused into the first conv's epilogue (Cout <= 128: one workgroup holds every cout of its
    voxels): conv + bias -> RMSNorm(C) * w -> x(1 + scale) + shift -> SiLU -> 16-bit (model/MBAblocks.py:196-203,356-367).
    Integer operands make the conv sum exact, so the only difference to the fp32 torch reference is the final rounding
    to the 16-bit type (half an ulp) plus fp32 rounding of the norm (1e-6 relative)."""
    g = torch.Generator().manual_seed(61)
    x = util.rand_int((N, Cin, 2, S, S), -3, 3, 61)
    w = util.rand_int((Cout, Cin, 3, 3, 3), -2, 2, 62)
    b = util.rand_int((Cout,), -4, 4, 63)
    nimg = (N + per_image - 1) // per_image
    nw = torch.rand(Cout, generator=g) + 0.5
    sc, sh = torch.randn((nimg, Cout), generator=g) * 0.3, torch.randn((nimg, Cout), generator=g) * 0.3
    v = F.conv3d(x, w, b, padding=1)
    img = torch.arange(N) // per_image
    ref = v * torch.rsqrt(v.pow(2).mean(1, keepdim=True) + 1e-6) * nw.view(1, -1, 1, 1, 1)
    ref = ref * (1 + sc[img].view(N, Cout, 1, 1, 1)) + sh[img].view(N, Cout, 1, 1, 1)
    ref = ref * torch.sigmoid(ref)
    got = util.conv27_fused(x.to(DEV), w, b, nw, sc, sh, per_image, dtype, waves).cpu()
    ulp = 2.0 ** -8 if dtype == "bf16" else 2.0 ** -11
    err = ((got - ref).abs() / (ref.abs() * ulp + 1e-4)).max().item()
    assert err <= 1.01, (err, util.report("fused " + dtype, got, ref))


# ---- windowed cross-attention core -------------------------------------------------------------------
def _window_attn_ref(q, k, v, qw, kw):
    """fp32 reference of model/MBAblocks.py:560-590 on NCDHW tensors (one head of width C, 2 x 2 windows)."""
    N, Cc, Z, S, _ = q.shape
    hs = S // 2

    def win(t):          # [N, C, Z, S, S] -> [N, 4, Z*hs*hs, C]
        t = t.reshape(N, Cc, Z, 2, hs, 2, hs).permute(0, 3, 5, 2, 4, 6, 1)
        return t.reshape(N, 4, Z * hs * hs, Cc)

    def rms(t, w):
        return t * torch.rsqrt(t.pow(2).mean(-1, keepdim=True) + 1e-6) * w

    qn, kn, vv = rms(win(q), qw), rms(win(k), kw), win(v)
    o = torch.softmax((qn / Cc) @ kn.transpose(-2, -1), -1) @ vv
    o = o.reshape(N, 2, 2, Z, hs, hs, Cc).permute(0, 6, 3, 1, 4, 2, 5)
    return o.reshape(N, Cc, Z, S, S)


@pytest.mark.parametrize("C_,S,Z,dtype", [(256, 16, 2, "f32"), (512, 8, 2, "f32"), (256, 16, 2, "bf16"), (512, 8, 2, "bf16"),
                                          (128, 16, 2, "bf16"), (256, 16, 1, "bf16"), (512, 8, 4, "bf16"),      # T = 64
                                          (256, 16, 4, "bf16"), (256, 16, 8, "bf16"), (128, 16, 8, "bf16"),     # T = 256, 512 (two-pass)
                                          (256, 16, 1, "f32"), (256, 16, 4, "f32"), (256, 16, 8, "f32"), (512, 4, 2, "f32"),
                                          (512, 8, 1, "f32")])                                               # generic: T = 64, 256, 512, 8, 16
def test_window_attention_core(C_, S, Z, dtype):
    N = 3
    g = torch.Generator().manual_seed(C_ + S)
    q, k, v = (torch.randn((N, C_, Z, S, S), generator=g) * s for s in (1.5, 0.7, 1.0))
    qw, kw = torch.rand(C_, generator=g) + 0.5, torch.rand(C_, generator=g) + 0.5
    k = k * 3 + q * 0.5                                   # structured logits: the softmax is far from uniform
    if dtype == "bf16":
        q, k, v = (t.bfloat16().float() for t in (q, k, v))
    ref = _window_attn_ref(q, k, v, qw, kw)
    qc, kc, vc = (util.to_cb8(t.to(DEV)) for t in (q, k, v))
    qwd, kwd = qw.to(DEV), kw.to(DEV)
    if dtype == "f32":
        out = torch.zeros_like(qc)
    else:
        out = torch.zeros(qc.shape, dtype=torch.bfloat16, device=DEV)
    _lib.check(_lib.lib().tm_op_window_attn(_lib.ptr(qc), _lib.ptr(kc), _lib.ptr(vc), _lib.ptr(qwd), _lib.ptr(kwd), _lib.ptr(out),
                                            N, C_, Z, S, 0 if dtype == "f32" else 1, _lib.current_stream_ptr()), "tm_op_window_attn")
    got = util.from_cb8(out.float(), C_).cpu()
    err = (got - ref).abs().max().item()
    # bf16: operands q*w, k, P, V rounded to 8 significant bits, fp32 accumulation and softmax
    tol = 2e-5 if dtype == "f32" else 3e-2
    assert err < tol, (err, util.report("window_attn", got, ref))
    if dtype == "bf16":
        assert ((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item() < 1e-2
