"""GPU suite (-m gpu): the HIP path, called through the C-ABI, against the CPU oracle and the golden fixtures.

Tolerances (stated, not tuned per case): the engine stores activations/weights in bf16 and accumulates in fp32, the
oracle is fp32 end to end, so single ops are held to 2e-2 of the reference's max magnitude (1e-2 when the oracle is fed
bf16-rounded operands), whole small networks to 3-4e-2 (rms 2e-2), few-step chains to 2-3e-2: 2-3x the measured errors
(tools/measure_small_net_errors.py prints them); integer outputs (labels) must be bit-exact whenever the
kernel is fed the same fp32 probabilities as the oracle.
"""
import json
import math
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import nets as O
from oracle import samplers as S
from test_oracle_golden import build_small, c_posterior
from util import AE_SMALL, CCDM_SMALL, LDM_SMALL, SEED, T, gold, rel_err, rms_err, sd_cpu, seeded

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from jointimagegeneration_amd import _lib
    _lib.load()                       # fail loudly if the HIP extension is missing
    return torch.device("cuda:0")


def bf(x):
    return x.to(torch.bfloat16).float()


# ------------------------------------------------------------------------------------------------ conv
CONV_CASES = [
    # name, dims, N, Cin, Cout, spatial, k, stride, pad, upsample
    ("3d_same", 3, 1, 64, 96, (5, 6, 7), 3, 1, 1, False),
    ("3d_stride2", 3, 2, 32, 64, (6, 8, 10), 3, 2, 1, False),
    ("3d_stride2_odd", 3, 1, 32, 32, (5, 7, 9), 3, 2, 1, False),
    ("3d_upsample", 3, 1, 64, 64, (3, 4, 5), 3, 1, 1, True),
    ("3d_stem_pad", 3, 1, 15, 64, (8, 8, 8), 3, 1, 1, False),
    ("3d_head", 3, 1, 64, 14, (8, 8, 8), 3, 1, 1, False),
    ("2d_same", 2, 3, 160, 320, (9, 11), 3, 1, 1, False),
    ("2d_wide", 2, 1, 1600, 800, (4, 4), 3, 1, 1, False),
    ("2d_stride2", 2, 2, 64, 64, (16, 16), 3, 2, 1, False),
    ("2d_ae_down", 2, 1, 32, 32, (16, 12), 3, 2, 0, False),
    ("2d_upsample", 2, 2, 96, 96, (7, 5), 3, 1, 1, True),
    ("2d_1x1", 2, 2, 192, 64, (8, 8), 1, 1, 0, False),
    ("1d_1x1_tokens", 1, 2, 64, 192, (50,), 1, 1, 0, False),
    ("2d_big_m", 2, 1, 32, 128, (70, 66), 3, 1, 1, False),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_matches_oracle(dev, case):
    from jointimagegeneration_amd import ops
    name, dims, N, Cin, Cout, sp, k, stride, pad, up = case
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000)
    x = torch.randn((N, Cin) + sp, generator=g)
    w = torch.randn((Cout, Cin) + (k,) * dims, generator=g) / math.sqrt(Cin * k ** dims)
    b = torch.randn(Cout, generator=g) * 0.1
    xin = O.upsample_nearest2(bf(x)) if up else bf(x)
    if stride == 2 and pad == 0:
        xin = F.pad(xin, (0, 1, 0, 1))
    ref = O.conv(xin, bf(w), b, stride=stride, padding=pad if not (stride == 2 and pad == 0) else 0)
    xcl = ops.to_cl(x.to(dev))
    pw = ops.pack_conv_weight(w.to(dev), xcl.Cpad)
    pb = ops.pad_bias(b.to(dev), Cout, dev)
    k3 = (1,) * (3 - dims) + (k,) * dims
    out = ops.conv(xcl, pw, pb, Cout, k=k3, stride=stride, pad=pad, upsample=up, out_f32=(Cout == 14))
    got = ops.from_cl(out, dims).cpu()
    assert got.shape == ref.shape
    assert rel_err(got, ref) < 1e-2, rel_err(got, ref)
    # pad lanes of the channels-last output must be exactly zero
    if out.Cpad > Cout:
        assert float(out.t[..., Cout:].float().abs().max()) == 0.0


HALO_CASES = [
    # name, dims, N, Cin, Cout, spatial(in), upsample   (3x3(x3), stride 1, pad 1; extents tile exactly: halo kernel)
    ("h3d_nt2", 3, 1, 64, 64, (8, 8, 32), False),
    ("h3d_nt4", 3, 2, 32, 128, (4, 8, 16), False),
    ("h3d_nt3_cin15", 3, 1, 15, 96, (8, 8, 16), False),
    ("h3d_head_f32", 3, 1, 64, 14, (4, 8, 16), False),
    ("h3d_up", 3, 1, 64, 64, (4, 4, 8), True),
    ("h2d_nt2", 2, 1, 160, 320, (32, 32), False),
    ("h2d_nt4", 2, 2, 96, 128, (32, 48), False),
    ("h2d_up", 2, 1, 128, 128, (16, 8), True),
    ("h2d_cout1", 2, 1, 128, 1, (32, 16), False),
]


@pytest.mark.parametrize("case", HALO_CASES, ids=[c[0] for c in HALO_CASES])
def test_conv_halo_kernel_matches_oracle(dev, case, halo_hint):
    """Same oracle, but shapes inside the halo-tile kernel's envelope (path_hint = 1 lifts the grid-fill gate)."""
    from jointimagegeneration_amd import ops
    name, dims, N, Cin, Cout, sp, up = case
    assert ops.conv_runs_halo_tile(ops.CL(torch.empty((N,) + (1,) * (3 - dims) + sp + (ops.pad32(Cin),), dtype=torch.bfloat16, device=dev), Cin),
                                   Cout, k=(1,) * (3 - dims) + (3,) * dims, upsample=up)        # really the halo kernel
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000)
    x = torch.randn((N, Cin) + sp, generator=g)
    w = torch.randn((Cout, Cin) + (3,) * dims, generator=g) / math.sqrt(Cin * 3 ** dims)
    b = torch.randn(Cout, generator=g) * 0.1
    ref = O.conv(O.upsample_nearest2(bf(x)) if up else bf(x), bf(w), b, padding=1)
    xcl = ops.to_cl(x.to(dev))
    pw = ops.pack_conv_weight(w.to(dev), xcl.Cpad)
    out = ops.conv(xcl, pw, ops.pad_bias(b.to(dev), Cout, dev), Cout, k=(1,) * (3 - dims) + (3,) * dims, upsample=up, out_f32=(Cout == 14))
    got = ops.from_cl(out, dims).cpu()
    assert got.shape == ref.shape
    assert rel_err(got, ref) < 1e-2, rel_err(got, ref)
    if out.Cpad > Cout:
        assert float(out.t[..., Cout:].float().abs().max()) == 0.0


def test_conv_halo_two_source_prologue_residual(dev, halo_hint):
    from jointimagegeneration_amd import ops
    g = torch.Generator().manual_seed(6)
    N, C1, C2, Cout, sp = 2, 64, 32, 64, (4, 8, 16)
    x1, x2 = torch.randn((N, C1) + sp, generator=g), torch.randn((N, C2) + sp, generator=g)
    w = torch.randn(Cout, C1 + C2, 3, 3, 3, generator=g) / math.sqrt((C1 + C2) * 27)
    tb = torch.randn(N, Cout, generator=g)
    res = torch.randn((N, Cout) + sp, generator=g)
    gamma, beta = 1 + 0.1 * torch.randn(C1 + C2, generator=g), 0.1 * torch.randn(C1 + C2, generator=g)
    xc = torch.cat([bf(x1), bf(x2)], 1)
    c1, c2 = ops.to_cl(x1.to(dev)), ops.to_cl(x2.to(dev))
    scale, shift = ops.groupnorm_stats(c1, gamma.to(dev), beta.to(dev), 1e-5, src2=c2)
    pw = ops.pack_conv_weight(w.to(dev), C1 + C2)
    tbp = torch.zeros(N, ops.pad32(Cout), device=dev); tbp[:, :Cout] = tb.to(dev)
    for silu in (True, False):
        a = O.group_norm(xc, gamma, beta, 1e-5)
        a = O.silu(a) if silu else a
        ref = O.conv(bf(a), bf(w), None, padding=1) + tb[:, :, None, None, None] + bf(res)
        out = ops.conv(c1, pw, tbp, Cout, k=(3, 3, 3), src2=c2, residual=ops.to_cl(res.to(dev)), bias_per_sample=True,
                       prologue=(scale, shift), prologue_silu=silu)
        assert rel_err(ops.from_cl(out, 3), ref) < 1.5e-2


def test_conv_two_source_residual_per_sample_bias_prologue(dev):
    from jointimagegeneration_amd import ops
    g = torch.Generator().manual_seed(5)
    N, C1, C2, Cout, sp = 2, 64, 32, 96, (4, 5, 6)
    x1, x2 = torch.randn((N, C1) + sp, generator=g), torch.randn((N, C2) + sp, generator=g)
    w = torch.randn(Cout, C1 + C2, 3, 3, 3, generator=g) / math.sqrt((C1 + C2) * 27)
    tb = torch.randn(N, Cout, generator=g)
    res = torch.randn((N, Cout) + sp, generator=g)
    gamma, beta = 1 + 0.1 * torch.randn(C1 + C2, generator=g), 0.1 * torch.randn(C1 + C2, generator=g)
    xc = torch.cat([bf(x1), bf(x2)], 1)
    a = O.silu(O.group_norm(xc, gamma, beta, 1e-5))
    ref = O.conv(bf(a), bf(w), None, padding=1) + tb[:, :, None, None, None] + bf(res)
    c1, c2 = ops.to_cl(x1.to(dev)), ops.to_cl(x2.to(dev))
    scale, shift = ops.groupnorm_stats(c1, gamma.to(dev), beta.to(dev), 1e-5, src2=c2)
    pw = ops.pack_conv_weight(w.to(dev), C1 + C2)
    tbp = torch.zeros(N, ops.pad32(Cout), device=dev); tbp[:, :Cout] = tb.to(dev)
    # (a) unfused: apply kernel then conv
    acl = ops.groupnorm_apply(c1, scale, shift, True, src2=c2)
    out = ops.conv(acl, pw, tbp, Cout, k=(3, 3, 3), residual=ops.to_cl(res.to(dev)), bias_per_sample=True)
    assert rel_err(ops.from_cl(out, 3), ref) < 1.5e-2
    # (b) fused GroupNorm*SiLU prologue + fused concat
    out2 = ops.conv(c1, pw, tbp, Cout, k=(3, 3, 3), src2=c2, residual=ops.to_cl(res.to(dev)), bias_per_sample=True,
                    prologue=(scale, shift))
    assert rel_err(ops.from_cl(out2, 3), ref) < 1.5e-2


def test_splitk_reduce_is_deterministic(dev):
    """Under-filled convs are split over K; the slab reduce runs in a fixed order: every launch gives the same bits."""
    from jointimagegeneration_amd import ops
    g = torch.Generator().manual_seed(12)
    for (N, Cin, Cout, sp) in ((1, 1600, 800, (4, 4)), (1, 320, 160, (20, 20)), (2, 640, 320, (12, 12))):
        x = ops.to_cl(torch.randn((N, Cin) + sp, generator=g).to(dev))
        w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
        res = ops.to_cl(torch.randn((N, Cout) + sp, generator=g).to(dev))
        pw, pb = ops.pack_conv_weight(w.to(dev), Cin), ops.pad_bias(torch.randn(Cout, generator=g).to(dev), Cout, dev)
        ref = ops.conv(x, pw, pb, Cout, k=(1, 3, 3), residual=res).t.clone()
        outs = [ops.conv(x, pw, pb, Cout, k=(1, 3, 3), residual=res).t for _ in range(50)]
        torch.cuda.synchronize()
        assert all(torch.equal(o, ref) for o in outs)


def test_conv_gather5_two_source_prologue_residual(dev):
    """160-channel-step gather variant (channel counts that are multiples of 160): fused concat, GN prologue, residual, split-K."""
    from jointimagegeneration_amd import ops
    g = torch.Generator().manual_seed(8)
    N, C1, C2, Cout, sp = 1, 320, 160, 160, (24, 8)           # W % 16 != 0: outside the halo / box kernels' envelope
    x1, x2 = torch.randn((N, C1) + sp, generator=g), torch.randn((N, C2) + sp, generator=g)
    w = torch.randn(Cout, C1 + C2, 3, 3, generator=g) / math.sqrt((C1 + C2) * 9)
    b = torch.randn(Cout, generator=g) * 0.1
    res = torch.randn((N, Cout) + sp, generator=g)
    gamma, beta = 1 + 0.1 * torch.randn(C1 + C2, generator=g), 0.1 * torch.randn(C1 + C2, generator=g)
    xc = torch.cat([bf(x1), bf(x2)], 1)
    c1, c2 = ops.to_cl(x1.to(dev)), ops.to_cl(x2.to(dev))
    scale, shift = ops.groupnorm_stats(c1, gamma.to(dev), beta.to(dev), 1e-5, src2=c2)
    pw, pb = ops.pack_conv_weight(w.to(dev), C1 + C2), ops.pad_bias(b.to(dev), Cout, dev)
    ref0 = O.conv(xc, bf(w), b, padding=1) + bf(res)
    out0 = ops.conv(c1, pw, pb, Cout, k=(1, 3, 3), src2=c2, residual=ops.to_cl(res.to(dev)))
    assert rel_err(ops.from_cl(out0, 2), ref0) < 1e-2
    ref1 = O.conv(bf(O.silu(O.group_norm(xc, gamma, beta, 1e-5))), bf(w), b, padding=1) + bf(res)
    out1 = ops.conv(c1, pw, pb, Cout, k=(1, 3, 3), src2=c2, residual=ops.to_cl(res.to(dev)), prologue=(scale, shift))
    assert rel_err(ops.from_cl(out1, 2), ref1) < 1.5e-2
    # stride 2 and 1x1 through the same variant
    w1 = torch.randn(320, 160, 1, 1, generator=g) / math.sqrt(160)
    xs = torch.randn(2, 160, 16, 16, generator=g)
    o = ops.conv(ops.to_cl(xs.to(dev)), ops.pack_conv_weight(w1.to(dev), 160), None, 320, k=(1, 1, 1), pad=0)
    assert rel_err(ops.from_cl(o, 2), O.conv(bf(xs), bf(w1))) < 1e-2
    w2 = torch.randn(160, 160, 3, 3, generator=g) / math.sqrt(160 * 9)
    o = ops.conv(ops.to_cl(xs.to(dev)), ops.pack_conv_weight(w2.to(dev), 160), None, 160, k=(1, 3, 3), stride=2, pad=1)
    assert rel_err(ops.from_cl(o, 2), O.conv(bf(xs), bf(w2), stride=2, padding=1)) < 1e-2


BOX_S2_CASES = [(1, 160, 160, (64, 64)), (1, 320, 320, (32, 32)), (1, 640, 640, (16, 16)), (1, 640, 640, (8, 8)), (2, 96, 64, (16, 16)),
                (1, 64, 40, (24, 16)), (1, 64, 64, (10, 8)), (3, 32, 96, (12, 32))]


@pytest.mark.parametrize("case", BOX_S2_CASES, ids=[f"n{c[0]}_{c[1]}to{c[2]}_{c[3][0]}x{c[3][1]}" for c in BOX_S2_CASES])
def test_conv_box_stride2_matches_oracle(dev, case):
    """UNet Downsample convs (3x3, stride 2, pad 1; unet.py Downsample / openaimodel.py:143-163) on the box-resident kernel: de-interleaved
    box columns, odd and ragged extents, batches, padded couts."""
    from jointimagegeneration_amd import ops
    N, Cin, Cout, sp = case
    g = torch.Generator().manual_seed(Cin + Cout + sp[0])
    x = torch.randn((N, Cin) + sp, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g) * 0.1
    ref = O.conv(bf(x), bf(w), b, stride=2, padding=1)
    xcl = ops.to_cl(x.to(dev))
    out = ops.conv(xcl, ops.pack_conv_weight(w.to(dev), xcl.Cpad), ops.pad_bias(b.to(dev), Cout, dev), Cout, k=(1, 3, 3), stride=2, pad=1)
    got = ops.from_cl(out, 2).cpu()
    assert got.shape == ref.shape
    assert rel_err(got, ref) < 1e-2, rel_err(got, ref)
    if out.Cpad > Cout:
        assert float(out.t[..., Cout:].float().abs().max()) == 0.0


BOX_CASES = [
    # name, N, Cin, Cout, spatial(in), upsample   (2-D 3x3 s1 p1, W % 16 == 0, H % 32 != 0: the halo kernel declines,
    # the box-resident kernel takes them; TH = rows of 16 positions per workgroup follows from the grid size)
    ("box_th2", 1, 160, 160, (16, 16), False),
    ("box_th4", 2, 32, 512, (20, 16), False),
    ("box_th8", 2, 64, 320, (48, 32), False),
    ("box_th2_up", 1, 96, 64, (8, 8), True),
    ("box_th8_up", 2, 32, 320, (24, 16), True),
    ("box_cout14_f32", 1, 64, 14, (16, 32), False),
    ("box_cin15", 1, 15, 64, (12, 16), False),
    # 8-wide and 4-wide position tiles (2x8, 4x4 positions per MFMA tile): the deepest UNet levels
    ("box_w8", 1, 640, 640, (8, 8), False),
    ("box_w8_n2", 2, 96, 64, (8, 8), False),
    ("box_w8_h24", 1, 64, 96, (24, 8), False),
    ("box_w4", 1, 800, 800, (4, 4), False),
    ("box_w4_n3", 3, 64, 32, (4, 4), False),
    ("box_w8_up", 2, 160, 160, (4, 4), True),
    ("box_w8_three_stages", 1, 1280, 64, (8, 8), False),
    # cout sub-split (2 / 4 workgroups per 16-cout tile, each taking its own weight rows): padded couts, batch 2, 8x8 level
    ("box_w4_sub4_cout40", 1, 800, 40, (4, 4), False),
    ("box_w4_sub2_n2", 2, 320, 800, (4, 4), False),
    ("box_w8_sub2_cout72", 1, 640, 72, (8, 8), False),
    # ragged row tiles (12 / 6 / 3 rows): 240 workgroups on the 64 / 32 / 16-row levels, last tile partly outside the image
    ("box_ragged12", 1, 160, 160, (64, 64), False),
    ("box_ragged6", 1, 64, 320, (32, 32), False),
    ("box_ragged3", 1, 96, 640, (16, 16), False),
    ("box_ragged_up", 1, 64, 320, (16, 16), True),
]


@pytest.mark.parametrize("case", BOX_CASES, ids=[c[0] for c in BOX_CASES])
def test_conv_box_kernel_matches_oracle(dev, case):
    from jointimagegeneration_amd import ops
    name, N, Cin, Cout, sp, up = case
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000)
    x = torch.randn((N, Cin) + sp, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g) * 0.1
    ref = O.conv(O.upsample_nearest2(bf(x)) if up else bf(x), bf(w), b, padding=1)
    xcl = ops.to_cl(x.to(dev))
    pw = ops.pack_conv_weight(w.to(dev), xcl.Cpad)
    out = ops.conv(xcl, pw, ops.pad_bias(b.to(dev), Cout, dev), Cout, k=(1, 3, 3), upsample=up, out_f32=(Cout == 14))
    got = ops.from_cl(out, 2).cpu()
    assert got.shape == ref.shape
    assert rel_err(got, ref) < 1e-2, rel_err(got, ref)
    if out.Cpad > Cout:
        assert float(out.t[..., Cout:].float().abs().max()) == 0.0


@pytest.mark.parametrize("cfg", [(1, 640, 1920, (8, 8)), (2, 160, 480, (16, 32)), (1, 800, 800, (4, 4)), (1, 320, 160, (24, 16))],
                         ids=["qkv_8x8", "qkv_16x32_n2", "proj_4x4", "skip_24x16"])
def test_conv_box_kernel_1x1_with_residual(dev, cfg):
    """1x1 convs (attention qkv / proj, ResBlock skip) on the box kernel: the box is the tile itself, one tap."""
    from jointimagegeneration_amd import ops
    N, Cin, Cout, sp = cfg
    g = torch.Generator().manual_seed(Cin + Cout)
    x = torch.randn((N, Cin) + sp, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / math.sqrt(Cin)
    b = torch.randn(Cout, generator=g) * 0.1
    res = torch.randn((N, Cout) + sp, generator=g)
    ref = O.conv(bf(x), bf(w), b) + bf(res)
    xcl = ops.to_cl(x.to(dev))
    out = ops.conv(xcl, ops.pack_conv_weight(w.to(dev), xcl.Cpad), ops.pad_bias(b.to(dev), Cout, dev), Cout, k=(1, 1, 1), pad=0,
                   residual=ops.to_cl(res.to(dev)))
    assert rel_err(ops.from_cl(out, 2), ref) < 1e-2


def test_conv_box_kernel_randomised_sweep(dev):
    """Seeded sweep over the box kernel's envelope (tile widths 16/8/4, ragged row tiles, 3x3 / 1x1, upsample, batch, two
    sources, prologue, residual, per-sample bias, Cout not a multiple of 32) against the oracle."""
    from jointimagegeneration_amd import ops
    rng = np.random.RandomState(1234)
    checked = 0
    for it in range(28):
        W = int(rng.choice([4, 8, 16, 32]))
        H = int(rng.choice([4, 6, 8, 12, 20, 24])) if W >= 8 else 4
        if H % 32 == 0:
            H += 4
        k = int(rng.choice([3, 3, 3, 1]))
        up = bool(k == 3 and rng.rand() < 0.25)
        if up and (H % 2 or W < 8):
            up = False
        N = int(rng.choice([1, 1, 2, 3]))
        C1 = int(rng.choice([32, 64, 96, 160]))
        C2 = int(rng.choice([0, 0, 32, 64])) if not up else 0
        Cout = int(rng.choice([14, 32, 48, 64, 160, 200]))
        use_pro = bool(rng.rand() < 0.4) and k == 3
        use_res = bool(rng.rand() < 0.5)
        per_sample = bool(rng.rand() < 0.5)
        hin, win = (H // 2, W // 2) if up else (H, W)
        g = torch.Generator().manual_seed(1000 + it)
        x1 = torch.randn(N, C1, hin, win, generator=g)
        x2 = torch.randn(N, C2, hin, win, generator=g) if C2 else None
        Cin = C1 + C2
        w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
        tb = torch.randn(N if per_sample else 1, Cout, generator=g) * 0.2
        res = torch.randn(N, Cout, H, W, generator=g) if use_res else None
        gamma, beta = 1 + 0.1 * torch.randn(Cin, generator=g), 0.1 * torch.randn(Cin, generator=g)
        xc = torch.cat([bf(x1), bf(x2)], 1) if C2 else bf(x1)
        c1 = ops.to_cl(x1.to(dev))
        c2 = ops.to_cl(x2.to(dev)) if C2 else None
        if not ops.conv_out_extent or c1.Cpad != C1:
            continue
        a = xc
        pro = None
        if use_pro:
            a = bf(O.silu(O.group_norm(xc, gamma, beta, 1e-5)))
            pro = ops.groupnorm_stats(c1, gamma.to(dev), beta.to(dev), 1e-5, src2=c2)
        a = O.upsample_nearest2(a) if up else a
        ref = O.conv(a, bf(w), None, padding=k // 2) + tb[:, :, None, None]
        if use_res:
            ref = ref + bf(res)
        tbp = torch.zeros(tb.shape[0], ops.pad32(Cout), device=dev); tbp[:, :Cout] = tb.to(dev)
        out = ops.conv(c1, ops.pack_conv_weight(w.to(dev), Cin), tbp if per_sample else tbp[0], Cout, k=(1, k, k), pad=k // 2, upsample=up,
                       src2=c2, residual=ops.to_cl(res.to(dev)) if use_res else None, bias_per_sample=per_sample, prologue=pro)
        err = rel_err(ops.from_cl(out, 2), ref)
        assert err < (1.5e-2 if use_pro else 1e-2), (it, N, C1, C2, Cout, H, W, k, up, use_pro, use_res, per_sample, err)
        if out.Cpad > Cout:
            assert float(out.t[..., Cout:].float().abs().max()) == 0.0
        checked += 1
    assert checked >= 20


def test_conv_box_two_source_prologue_residual_two_stages(dev):
    """Box kernel with everything fused: concat of two sources, GroupNorm(*SiLU) prologue, per-sample bias, residual; 1280 input
    channels do not fit one LDS stage, so the box is staged twice."""
    from jointimagegeneration_amd import ops
    g = torch.Generator().manual_seed(9)
    N, C1, C2, Cout, sp = 2, 640, 640, 64, (16, 16)
    x1, x2 = torch.randn((N, C1) + sp, generator=g), torch.randn((N, C2) + sp, generator=g)
    w = torch.randn(Cout, C1 + C2, 3, 3, generator=g) / math.sqrt((C1 + C2) * 9)
    tb = torch.randn(N, Cout, generator=g)
    res = torch.randn((N, Cout) + sp, generator=g)
    gamma, beta = 1 + 0.1 * torch.randn(C1 + C2, generator=g), 0.1 * torch.randn(C1 + C2, generator=g)
    xc = torch.cat([bf(x1), bf(x2)], 1)
    c1, c2 = ops.to_cl(x1.to(dev)), ops.to_cl(x2.to(dev))
    scale, shift = ops.groupnorm_stats(c1, gamma.to(dev), beta.to(dev), 1e-5, src2=c2)
    pw = ops.pack_conv_weight(w.to(dev), C1 + C2)
    tbp = torch.zeros(N, ops.pad32(Cout), device=dev); tbp[:, :Cout] = tb.to(dev)
    for silu in (True, False):
        a = O.group_norm(xc, gamma, beta, 1e-5)
        a = O.silu(a) if silu else a
        ref = O.conv(bf(a), bf(w), None, padding=1) + tb[:, :, None, None] + bf(res)
        out = ops.conv(c1, pw, tbp, Cout, k=(1, 3, 3), src2=c2, residual=ops.to_cl(res.to(dev)), bias_per_sample=True,
                       prologue=(scale, shift), prologue_silu=silu)
        assert rel_err(ops.from_cl(out, 2), ref) < 1.5e-2
    # determinism: the 4-wave combine has a fixed order
    o1 = ops.conv(c1, pw, tbp, Cout, k=(1, 3, 3), src2=c2, bias_per_sample=True).t
    o2 = ops.conv(c1, pw, tbp, Cout, k=(1, 3, 3), src2=c2, bias_per_sample=True).t
    assert torch.equal(o1, o2)


def test_conv_rejects_bad_shapes(dev):
    from jointimagegeneration_amd import ops
    x = ops.to_cl(torch.randn(1, 32, 4, 4, device=dev))
    pw = ops.pack_conv_weight(torch.randn(32, 32, 5, 5, device=dev), 32)
    with pytest.raises(RuntimeError, match="kernel extent"):
        ops.conv(x, pw, None, 32, k=(1, 5, 5), pad=1)


# ------------------------------------------------------------------------------------------------ norms / small ops
@pytest.mark.parametrize("shape", [(2, 64, 3, 5, 7), (1, 320, 16, 16), (1, 192, 20, 20, 20), (3, 1600, 4, 4)])
def test_groupnorm_silu(dev, shape):
    from jointimagegeneration_amd import ops
    g = torch.Generator().manual_seed(shape[1])
    x = torch.randn(shape, generator=g) * 2 + 0.5
    C = shape[1]
    gamma, beta = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    for eps, act in ((1e-5, True), (1e-6, False)):
        ref = O.group_norm(bf(x), gamma, beta, eps)
        ref = O.silu(ref) if act else ref
        cl = ops.to_cl(x.to(dev))
        sc, sh = ops.groupnorm_stats(cl, gamma.to(dev), beta.to(dev), eps)
        got = ops.from_cl(ops.groupnorm_apply(cl, sc, sh, act), len(shape) - 2)
        assert rel_err(got, ref) < 1e-2


@pytest.mark.parametrize("cfg", [(1, 160, 160, (16, 16), 3), (2, 320, 640, (8, 8), 3), (1, 160, 320, (32, 32), 1), (1, 640, 640, (8, 8), 1),
                                 (1, 800, 800, (4, 4), 3), (1, 160, 160, (44, 32), 3), (1, 64, 320, (20, 16), 3)],
                         ids=["box3x3_16", "box3x3_8_n2", "gather5_1x1_32", "box1x1_8", "box3x3_4", "box_44x32_ragged", "box_20x16_ragged"])
def test_conv_epilogue_groupnorm_statistics(dev, cfg, monkeypatch):
    """Convs of the latent UNet leave per-channel fixed-point (sum, sumsq) of their bf16 outputs behind (gg_conv_desc.gn_acc);
    gg_groupnorm_apply_acc normalises from them.  Checked against the stored tensor's own statistics, against the oracle's
    GroupNorm, and for bit-reproducibility (integer atomics commute)."""
    from jointimagegeneration_amd import ops
    monkeypatch.setattr(ops, "GN_ACC_MIN_ELEMS", 0)        # production only asks for the sums where the statistics launch is slow
    N, Cin, Cout, sp, k = cfg
    g = torch.Generator().manual_seed(Cin + Cout + k)
    x = torch.randn((N, Cin) + sp, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=g) * 0.1
    res = torch.randn((N, Cout) + sp, generator=g)
    gamma, beta = 1 + 0.1 * torch.randn(Cout, generator=g), 0.1 * torch.randn(Cout, generator=g)
    xcl = ops.to_cl(x.to(dev))
    pw, pb = ops.pack_conv_weight(w.to(dev), xcl.Cpad), ops.pad_bias(b.to(dev), Cout, dev)
    outs = []
    for _ in range(2):
        ops.stats_begin(dev)
        y = ops.conv(xcl, pw, pb, Cout, k=(1, k, k), pad=k // 2, residual=ops.to_cl(res.to(dev)))
        ops.stats_end(dev)
        assert y.acc is not None, "this shape is expected to emit statistics"
        outs.append((y, y.acc.clone()))
    assert torch.equal(outs[0][1], outs[1][1])                             # bit-reproducible
    y, acc = outs[1]
    yt = y.t.float().reshape(N, -1, y.Cpad)                                # [N, S, C]
    s_ref, q_ref = yt.sum(1).double(), (yt * yt).sum(1).double()
    s_got, q_got = acc.sum(1)[..., 0].double() / 2 ** 28, acc.sum(1)[..., 1].double() / 2 ** 20
    S = yt.shape[1]
    assert float((s_got - s_ref).abs().max()) < 1e-3 * S ** 0.5 + 1e-3
    assert float(((q_got - q_ref).abs() / (q_ref.abs() + 1.0)).max()) < 1e-3
    ref = O.silu(O.group_norm(ops.from_cl(y, 2).cpu(), gamma, beta, 1e-5))
    got = ops.groupnorm_apply_acc(ops.CL(y.t, y.C, acc), gamma.to(dev), beta.to(dev), 1e-5, True)
    assert rel_err(ops.from_cl(got, 2), ref) < 1e-2
    sc, sh = ops.groupnorm_stats(y, gamma.to(dev), beta.to(dev), 1e-5)
    old = ops.groupnorm_apply(y, sc, sh, True)
    assert float((got.t.float() - old.t.float()).abs().max()) <= 2e-2 * float(ref.abs().max())
    # two sources (skip concat): statistics of both tensors, groups straddling the boundary
    if cfg[0] == 1 and Cout == 160:
        y2 = outs[0][0]
        gamma2, beta2 = 1 + 0.1 * torch.randn(320, generator=g), 0.1 * torch.randn(320, generator=g)
        cat = torch.cat([ops.from_cl(y, 2).cpu(), ops.from_cl(y2, 2).cpu()], 1)
        ref2 = O.group_norm(cat, gamma2, beta2, 1e-6)
        got2 = ops.groupnorm_apply_acc(ops.CL(y.t, y.C, acc), gamma2.to(dev), beta2.to(dev), 1e-6, False, src2=ops.CL(y2.t, y2.C, outs[0][1]))
        assert rel_err(ops.from_cl(got2, 2), ref2) < 1e-2


@pytest.mark.parametrize("shape", [(2, 640, 8, 8), (1, 800, 4, 4), (1, 1440, 8, 8), (1, 320, 8, 16), (3, 160, 7, 9), (1, 1600, 4, 4)])
def test_groupnorm_fused_single_launch(dev, shape):
    """One-launch statistics + apply for small tensors == oracle GroupNorm(+SiLU), and == the two-launch path up to bf16 rounding;
    groups whose channel range shares 16-byte pieces with the neighbours (cpg = 5, 25, 45, 50) and the two-source concat."""
    from jointimagegeneration_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=g) * 1.7 + 0.3
    Cc = shape[1]
    gamma, beta = 1 + 0.1 * torch.randn(Cc, generator=g), 0.1 * torch.randn(Cc, generator=g)
    xcl = ops.to_cl(x.to(dev))
    assert ops.groupnorm_fused_ok(xcl)
    for act in (True, False):
        ref = O.group_norm(bf(x), gamma, beta, 1e-5)
        ref = O.silu(ref) if act else ref
        got = ops.from_cl(ops.groupnorm_fused(xcl, gamma.to(dev), beta.to(dev), 1e-5, act), 2).cpu()
        assert rel_err(got, ref) < 1e-2
        sc, sh = ops.groupnorm_stats(xcl, gamma.to(dev), beta.to(dev), 1e-5)
        two = ops.from_cl(ops.groupnorm_apply(xcl, sc, sh, act), 2).cpu()
        assert float((got - two).abs().max()) <= 2.0 ** -6 * float(ref.abs().max())       # at most a bf16 ulp or two apart
    if Cc % 64 == 0:                                                                        # two sources: cat[x[:, :C/2], x[:, C/2:]]
        a, b2 = ops.to_cl(x[:, :Cc // 2].contiguous().to(dev)), ops.to_cl(x[:, Cc // 2:].contiguous().to(dev))
        assert ops.groupnorm_fused_ok(a, b2)
        got2 = ops.from_cl(ops.groupnorm_fused(a, gamma.to(dev), beta.to(dev), 1e-5, True, b2), 2).cpu()
        assert torch.equal(got2, ops.from_cl(ops.groupnorm_fused(xcl, gamma.to(dev), beta.to(dev), 1e-5, True), 2).cpu())
    big = ops.CL(torch.zeros(1, 1, 64, 64, 320, dtype=torch.bfloat16, device=dev), 320)
    assert not ops.groupnorm_fused_ok(big)


@pytest.mark.parametrize("shape", [(1, 640, 8, 8), (2, 320, 16, 16), (1, 160, 64, 64)], ids=["fused_small", "stats_small", "acc_64x64"])
def test_groupnorm_large_mean_to_std_ratio_all_paths(dev, shape):
    """|mean| = 200 x std (var = E[x^2] - mean^2 amplifies any error of 1/count by mean^2/var = 4e4): every GroupNorm
    path -- two-launch / small-tensor statistics (fp32 scale, shift tables), one-launch fused, accumulator-fed apply, accumulator
    fold -- against an fp64 GroupNorm of the same bf16 values."""
    from jointimagegeneration_amd import ops
    N, Cc = shape[:2]
    g = torch.Generator().manual_seed(Cc)
    x = bf(torch.randn(shape, generator=g) + 200.0)
    gamma, beta = 1 + 0.1 * torch.randn(Cc, generator=g), 0.1 * torch.randn(Cc, generator=g)
    xg = x.double().view(N, 32, -1)
    mean, var = xg.mean(-1), xg.var(-1, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    cpg = Cc // 32
    sc_ref = (rstd.repeat_interleave(cpg, 1) * gamma.double())
    sh_ref = beta.double() - mean.repeat_interleave(cpg, 1) * sc_ref
    ref = O.silu((x.double() * sc_ref.view(N, Cc, 1, 1) + sh_ref.view(N, Cc, 1, 1)).float())
    xcl = ops.to_cl(x.to(dev))
    sc, sh = ops.groupnorm_stats(xcl, gamma.to(dev), beta.to(dev), 1e-5)
    assert float(((sc.cpu().double() - sc_ref) / sc_ref).abs().max()) < 2e-5
    assert float(((sh.cpu().double() - sh_ref) / sh_ref).abs().max()) < 2e-5
    outs = {"two_launch": ops.groupnorm_apply(xcl, sc, sh, True)}
    if ops.groupnorm_fused_ok(xcl):
        outs["fused"] = ops.groupnorm_fused(xcl, gamma.to(dev), beta.to(dev), 1e-5, True)
    # accumulators as a producing conv would leave them: exact fixed-point per-channel sums of the stored bf16 values
    xt = xcl.t.double().reshape(N, -1, xcl.Cpad)
    acc = torch.stack([(xt.sum(1) * 2.0 ** 28).round().long(), ((xt * xt).sum(1) * 2.0 ** 20).round().long()], -1).view(N, 1, xcl.Cpad, 2).contiguous()
    outs["apply_acc"] = ops.groupnorm_apply_acc(ops.CL(xcl.t, Cc, acc), gamma.to(dev), beta.to(dev), 1e-5, True)
    sc2, sh2 = ops.groupnorm_scale_shift_acc(ops.CL(xcl.t, Cc, acc), gamma.to(dev), beta.to(dev), 1e-5)
    assert float(((sc2.cpu().double() - sc_ref) / sc_ref).abs().max()) < 2e-5
    for name, o in outs.items():
        got = ops.from_cl(o, 2).cpu()
        # the normalised values are differences of two numbers of size 200 * scale: an fp32 affine leaves ~200 * 2^-24 * scale ~ 1e-5
        err = float((got - ref).abs().max())
        assert err < 2.0 ** -7 * float(ref.abs().max()), (name, err)


def test_halo_conv_epilogue_statistics_replace_the_stats_pass(dev, halo_hint):
    """The halo-tile conv emits, per output channel, the exact fixed-point sum / sum of squares of its bf16-rounded outputs (32
    stripes); gg_groupnorm_scale_shift_acc folds them into the same per-(n, c) scale / shift the statistics PASS computes, for one
    source and for the skip concat (two producers, one of them with a residual epilogue), channel padding included."""
    from jointimagegeneration_amd import ops
    g = torch.Generator().manual_seed(77)
    N, sp = 2, (4, 8, 32)
    x = ops.to_cl(torch.randn((N, 64) + sp, generator=g).to(dev))
    res = ops.to_cl(torch.randn((N, 96) + sp, generator=g).to(dev))
    w1 = torch.randn(96, 64, 3, 3, 3, generator=g).to(dev) / math.sqrt(64 * 27)
    w2 = torch.randn(64, 64, 3, 3, 3, generator=g).to(dev) / math.sqrt(64 * 27)
    b1 = ops.pad_bias(torch.randn(96, generator=g).to(dev), 96, dev)
    ops.stats_begin(dev)
    try:
        y1 = ops.conv(x, ops.pack_conv_weight(w1, 64), b1, 96, k=(3, 3, 3), residual=res)
        y2 = ops.conv(x, ops.pack_conv_weight(w2, 64), None, 64, k=(3, 3, 3))
    finally:
        ops.stats_end(dev)
    assert y1.acc is not None and tuple(y1.acc.shape) == (N, 32, 96, 2) and tuple(y2.acc.shape) == (N, 32, 64, 2)
    # exact integer sums: compare with an fp64 sum of the stored bf16 values
    s64 = y1.t.double().sum((1, 2, 3))                                                  # [N, 96]
    got = y1.acc.sum(1)[..., 0].double() / 2.0 ** 28
    assert float((got - s64).abs().max()) < 1e-3
    q64 = (y1.t.double() ** 2).sum((1, 2, 3))
    gotq = y1.acc.sum(1)[..., 1].double() / 2.0 ** 20
    assert float(((gotq - q64).abs() / q64.abs().clamp_min(1.0)).max()) < 1e-4
    gam, bet = (1 + 0.1 * torch.randn(160, generator=g)).to(dev), (0.1 * torch.randn(160, generator=g)).to(dev)
    for a, b2, C in ((y1, None, 96), (y1, y2, 160), (y2, None, 64)):
        sc_ref, sh_ref = ops.groupnorm_stats(a, gam[:C].contiguous(), bet[:C].contiguous(), 1e-5, b2)
        sc, sh = ops.groupnorm_scale_shift_acc(a, gam[:C].contiguous(), bet[:C].contiguous(), 1e-5, b2)
        assert sc.shape == sc_ref.shape
        assert float((sc - sc_ref).abs().max()) < 2e-5 * float(sc_ref.abs().max()) + 1e-6
        assert float((sh - sh_ref).abs().max()) < 2e-5 * float(sh_ref.abs().max()) + 1e-5


def test_layernorm_geglu_add_linear_embedding(dev):
    from jointimagegeneration_amd import ops
    g = torch.Generator().manual_seed(9)
    x = torch.randn(3, 50, 320, generator=g)
    gamma, beta = 1 + 0.1 * torch.randn(320, generator=g), 0.1 * torch.randn(320, generator=g)
    got = ops.layernorm(x.to(dev).bfloat16(), gamma.to(dev), beta.to(dev))
    assert rel_err(got, F.layer_norm(bf(x), (320,), gamma, beta, 1e-5)) < 1e-2
    h = torch.randn(2, 20, 2 * 256, generator=g)
    a, gate = bf(h).chunk(2, dim=-1)
    assert rel_err(ops.geglu(h.to(dev).bfloat16(), 256), a * F.gelu(gate)) < 1e-2
    y = torch.randn(2, 20, 256, generator=g)
    z = torch.randn(2, 20, 256, generator=g)
    assert rel_err(ops.add(y.to(dev).bfloat16(), z.to(dev).bfloat16()), bf(y) + bf(z)) < 1e-2
    W, b, e = torch.randn(96, 128, generator=g) / 11, torch.randn(96, generator=g), torch.randn(5, 128, generator=g)
    got = ops.linear_f32(e.to(dev), W.to(dev), b.to(dev), act_in=True)
    assert rel_err(got, F.linear(O.silu(e), W, b)) < 1e-5
    tg = gold("timestep_embedding")
    assert rel_err(ops.timestep_embedding(T(tg["t_f"]).to(dev), 64), T(tg["emb_f"])) < 2e-5
    assert rel_err(ops.timestep_embedding(T(tg["t_i"]).float().to(dev), 160), T(tg["emb_i"])) < 2e-4


@pytest.mark.parametrize("shape", [(2, 50, 320, 1280), (1, 256, 64, 256), (1, 7, 96, 48), (3, 1024, 160, 640)], ids=lambda c: "N%dT%dC%di%d" % c)
def test_feed_forward_projection_with_geglu_epilogue(dev, shape, monkeypatch):
    """GEGLU as the epilogue of the feed-forward projection (gg_conv_desc.epilogue_geglu; attention.py:37-44): == value * gelu(gate) of
    the fp32 projection of the bf16-rounded operands, and == the two-launch path (projection rounded to bf16, then gg_geglu) up to
    that rounding; the whole BasicTransformerBlock gives the same output either way within a bf16 ulp or two."""
    from jointimagegeneration_amd import blocks as B
    from jointimagegeneration_amd import ops
    N, Tn, Cc, inner = shape
    g = torch.Generator().manual_seed(Tn + inner)
    x = torch.randn(N, Tn, Cc, generator=g)
    proj = torch.nn.Linear(Cc, 2 * inner)
    with torch.no_grad():
        proj.weight.copy_(torch.randn(2 * inner, Cc, generator=g) / math.sqrt(Cc)); proj.bias.copy_(0.3 * torch.randn(2 * inner, generator=g))
    proj = proj.to(dev)
    h = F.linear(bf(x), bf(proj.weight.cpu()), proj.bias.cpu())
    a, gate = h.chunk(2, dim=-1)
    ref = a * F.gelu(gate)
    xcl = ops.CL(torch.zeros(N, 1, 1, Tn, ops.pad32(Cc), dtype=torch.bfloat16, device=dev), Cc)
    xcl.t[..., :Cc] = x.to(dev).view(N, 1, 1, Tn, Cc)
    pw, pb = B.packed_geglu(proj, xcl.Cpad)
    got = ops.conv(xcl, pw, pb, 2 * inner, k=(1, 1, 1), pad=0, geglu=True)
    assert got.C == inner and got.t.shape[-1] == ops.pad32(2 * inner) // 2
    assert rel_err(got.t[..., :inner].float().view(N, Tn, inner).cpu(), ref) < 6e-3
    pw0, pb0 = B.packed_conv(proj, xcl.Cpad)
    two = ops.geglu(ops.conv(xcl, pw0, pb0, 2 * inner, k=(1, 1, 1), pad=0).t, inner)
    assert rel_err(got.t[..., :inner].float(), two.float()) < 1.5e-2          # the two-launch path rounds the projection to bf16 first
    with pytest.raises(ValueError):
        ops.conv(xcl, pw, pb, 2 * inner, k=(1, 1, 1), pad=0, geglu=True, residual=got)
    if Cc == 64:
        blk = seeded(B.BasicTransformerBlock(64, 2, 32, context_dim=48), "geglu_blk.").to(dev)
        xin = ops.CL((0.5 * torch.randn(1, 1, 1, Tn, 64, generator=g)).to(dev).bfloat16(), 64)
        ctx = ops.CL(torch.zeros(1, 1, 1, 9, 64, dtype=torch.bfloat16, device=dev), 48)
        ctx.t[..., :48] = torch.randn(1, 1, 1, 9, 48, generator=g).to(dev)
        fused = blk.run(xin, ctx).t.float()
        monkeypatch.setattr(B, "FUSE_GEGLU", False)
        plain = blk.run(xin, ctx).t.float()
        assert float((fused - plain).abs().max()) <= 2.0 ** -6 * float(plain.abs().max())


# ------------------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("cfg", [(2, 2, 32, 128), (1, 8, 32, 512), (2, 3, 32, 64), (1, 1, 64, 64), (1, 1, 512, 96), (1, 1, 384, 40), (1, 4, 128, 70),
                                 (1, 10, 32, 1024), (1, 3, 32, 1100), (1, 2, 64, 300), (1, 1, 128, 333)],      # (in-workgroup key split: T >= 4 key tiles, under-filled grid; ragged halves)
                         ids=lambda c: f"N{c[0]}h{c[1]}d{c[2]}T{c[3]}")
def test_attention_legacy_layout(dev, cfg):
    from jointimagegeneration_amd import ops
    N, heads, ch, Tn = cfg
    g = torch.Generator().manual_seed(Tn)
    qkv = torch.randn(N, heads * 3 * ch, Tn, generator=g)
    ref = O.qkv_attention_legacy(bf(qkv), heads)                     # [N, heads*ch, T]
    qcl = qkv.permute(0, 2, 1).contiguous().to(dev).bfloat16()       # [N, T, 3C] channels-last
    out = torch.empty(N, Tn, heads * ch, dtype=torch.bfloat16, device=dev)
    ld = heads * 3 * ch
    ops.attention(qcl, qcl, qcl, out, N, heads, ch, Tn, Tn, (ld, 3 * ch), (ld, 3 * ch), (ld, 3 * ch), (heads * ch, ch),
                  1.0 / math.sqrt(ch), q_off=0, k_off=ch, v_off=2 * ch)
    assert rel_err(out.float().permute(0, 2, 1), ref) < 2e-2


@pytest.mark.parametrize("cfg", [(1, 1, 512, 4096), (1, 1, 384, 1030), (2, 1, 256, 777)], ids=lambda c: f"N{c[0]}h{c[1]}d{c[2]}T{c[3]}")
def test_attention_key_split_single_head(dev, cfg):
    """Under-filled single-head grids (the AE mid-block attention: 64 workgroups at T = 4096) split the KEYS over workgroups and merge the
    online-softmax states in a second launch (gg_attention_workspace_bytes > 0): same result as the oracle and as the unsplit kernel
    (no workspace), ragged key ranges and empty trailing ranges included."""
    import ctypes as C
    from jointimagegeneration_amd import _lib, ops
    N, heads, ch, Tn = cfg
    g = torch.Generator().manual_seed(Tn)
    qkv = torch.randn(N, heads * 3 * ch, Tn, generator=g) * 0.5
    ref = O.qkv_attention_legacy(bf(qkv), heads)
    qcl = qkv.permute(0, 2, 1).contiguous().to(dev).bfloat16()
    ld = heads * 3 * ch
    d = _lib.AttentionDesc()
    d.N, d.heads, d.head_dim, d.Tq, d.Tkv = N, heads, ch, Tn, Tn
    d.ldq = d.ldk = d.ldv = ld
    d.hsq = d.hsk = d.hsv = 3 * ch
    d.ldo, d.hso, d.scale = heads * ch, ch, 1.0 / math.sqrt(ch)
    assert _lib.load().gg_attention_workspace_bytes(C.byref(d)) > 0
    out = torch.empty(N, Tn, heads * ch, dtype=torch.bfloat16, device=dev)
    ops.attention(qcl, qcl, qcl, out, N, heads, ch, Tn, Tn, (ld, 3 * ch), (ld, 3 * ch), (ld, 3 * ch), (heads * ch, ch), 1.0 / math.sqrt(ch),
                  q_off=0, k_off=ch, v_off=2 * ch)
    assert rel_err(out.float().permute(0, 2, 1), ref) < 2e-2
    # unsplit (no workspace handed in): the same numbers up to the order of the fp32 state merges
    d.q, d.k, d.v = qcl.data_ptr(), qcl.data_ptr() + 2 * ch, qcl.data_ptr() + 4 * ch
    out1 = torch.empty_like(out)
    d.out = out1.data_ptr()
    _lib.check(_lib.load().gg_attention_forward(C.byref(d), torch.cuda.current_stream().cuda_stream), "gg_attention_forward")
    assert float((out.float() - out1.float()).abs().max()) <= 2.0 ** -6 * float(out1.float().abs().max())


def test_attention_cross_context(dev):
    from jointimagegeneration_amd import ops
    g = torch.Generator().manual_seed(3)
    N, heads, d, Tq, L = 2, 2, 32, 64, 7
    q, k, v = torch.randn(N, Tq, heads * d, generator=g), torch.randn(N, L, heads * d, generator=g), torch.randn(N, L, heads * d, generator=g)

    def split(t):
        return bf(t).reshape(N, -1, heads, d).permute(0, 2, 1, 3)
    sim = torch.einsum("bhid,bhjd->bhij", split(q), split(k)) * d ** -0.5
    ref = torch.einsum("bhij,bhjd->bhid", sim.softmax(-1), split(v)).permute(0, 2, 1, 3).reshape(N, Tq, heads * d)
    kv = torch.cat([k, v], -1).to(dev).bfloat16().contiguous()
    out = torch.empty(N, Tq, heads * d, dtype=torch.bfloat16, device=dev)
    ops.attention(q.to(dev).bfloat16().contiguous(), kv, kv, out, N, heads, d, Tq, L, (heads * d, d), (2 * heads * d, d), (2 * heads * d, d),
                  (heads * d, d), d ** -0.5, k_off=0, v_off=heads * d)
    assert rel_err(out, ref) < 2e-2


# ------------------------------------------------------------------------------------------------ samplers
def test_posterior_kernel_bit_exact_labels(dev):
    """Same fp32 probabilities + same exponential tape => labels identical to the plain-C oracle AND to the reference."""
    from jointimagegeneration_amd import ops
    g = gold("ccdm_posterior")
    K = 14
    for t in (1, 2, 25, 50):
        lab, p0, E = T(g[f"t{t}_xt_labels"]).int().reshape(-1), T(g[f"t{t}_p0"]), T(g[f"t{t}_E"])
        a, abar = [float(v) for v in g[f"t{t}_a_abar"]]
        p0_cl = p0.permute(0, 2, 3, 4, 1).reshape(-1, K).contiguous()
        sc = torch.tensor([a, abar], dtype=torch.float32, device=dev)
        probs = torch.empty(p0_cl.shape, device=dev)
        got = ops.ccdm_posterior_sample(p0_cl.to(dev), False, lab.to(dev), sc, K, E=E.to(dev), draw=True, probs_out=probs)
        labc, prc = c_posterior(p0_cl, lab, E, a, abar, K)
        assert torch.equal(got.cpu(), labc)
        assert torch.equal(got.cpu(), T(g[f"t{t}_sample_labels"]).reshape(-1))          # the reference's own draw
        assert torch.equal(probs.cpu(), prc)                                             # fp32 posterior bit-identical to C
    # larger random problem, every K from 2..20, incl. argmax mode (E = None)
    gen = torch.Generator().manual_seed(77)
    for K in (2, 3, 12, 14, 16, 20):
        M = 5000
        p0 = torch.softmax(3 * torch.randn(M, K, generator=gen), -1)
        lab = torch.randint(0, K, (M,), generator=gen).int()
        E = torch.empty(M, K).exponential_(1, generator=gen)
        a, abar = 0.93, 0.41
        sc = torch.tensor([a, abar], dtype=torch.float32, device=dev)
        oh = torch.zeros(M, 32, dtype=torch.bfloat16, device=dev)
        got = ops.ccdm_posterior_sample(p0.to(dev), False, lab.to(dev), sc, K, E=E.to(dev), draw=True, onehot_out=oh)
        labc, _ = c_posterior(p0, lab, E, a, abar, K)
        assert torch.equal(got.cpu(), labc)
        assert torch.equal(oh[:, :K].float().argmax(-1).cpu().int(), labc) and float(oh[:, :K].float().sum()) == M
        got0 = ops.ccdm_posterior_sample(p0.to(dev), False, lab.to(dev), sc, K, draw=False)
        lab0, _ = c_posterior(p0, lab, None, a, abar, K)
        assert torch.equal(got0.cpu(), lab0)


def test_posterior_kernel_logits_and_philox(dev):
    from jointimagegeneration_amd import ops
    gen = torch.Generator().manual_seed(78)
    K, M = 14, 20000
    logits = 2 * torch.randn(M, 16, generator=gen)
    lab = torch.randint(0, K, (M,), generator=gen).int()
    E = torch.empty(M, K).exponential_(1, generator=gen)
    sc = torch.tensor([0.9, 0.5], dtype=torch.float32, device=dev)
    probs = torch.empty(M, K, device=dev)
    got = ops.ccdm_posterior_sample(logits.to(dev), True, lab.to(dev), sc, K, E=E.to(dev), draw=True, probs_out=probs)
    labc, prc = c_posterior(torch.softmax(logits[:, :K], -1), lab, E, 0.9, 0.5, K)
    # expf differs in the last ulp between glibc and the GPU: labels may differ only at near-ties of the race
    assert int((got.cpu() != labc).sum()) <= 2
    assert torch.allclose(probs.cpu(), prc, rtol=1e-5, atol=1e-9)
    # Philox path: empirical label frequencies follow the posterior (chi-square-ish bound), deterministic per (seed, offset)
    p0 = torch.softmax(torch.randn(1, K, generator=gen), -1).repeat(200000, 1)
    lab = torch.zeros(200000, dtype=torch.int32)
    off = torch.tensor([7], dtype=torch.int64, device=dev)
    pr = torch.empty(200000, K, device=dev)
    a = ops.ccdm_posterior_sample(p0.to(dev), False, lab.to(dev), sc, K, philox_seed=1234, philox_offset=off, draw=True, probs_out=pr)
    b = ops.ccdm_posterior_sample(p0.to(dev), False, lab.to(dev), sc, K, philox_seed=1234, philox_offset=off, draw=True)
    assert torch.equal(a, b)
    freq = torch.bincount(a.cpu().long(), minlength=K).float() / 200000
    assert float((freq - pr[0].cpu()).abs().max()) < 5e-3
    off2 = torch.tensor([8], dtype=torch.int64, device=dev)
    c = ops.ccdm_posterior_sample(p0.to(dev), False, lab.to(dev), sc, K, philox_seed=1234, philox_offset=off2, draw=True)
    expect = 1.0 - float((pr[0] ** 2).sum())                  # two independent draws differ with probability 1 - sum p^2
    assert abs(float((a != c).float().mean()) - expect) < 1e-2


def test_ddim_step_and_minmax(dev):
    from jointimagegeneration_amd import ops
    gen = torch.Generator().manual_seed(4)
    x, e, nz = torch.randn(2, 4, 8, 8, generator=gen), torch.randn(2, 4, 8, 8, generator=gen), torch.randn(2, 4, 8, 8, generator=gen)
    sch = S.ddim_schedule(T(gold("schedules")["ldm_alphas_cumprod"]), 50, eta=0.5)
    i = 20
    ref, ref0 = S.ddim_step(x, e, sch["alphas"][i], sch["alphas_prev"][i], sch["sigmas"][i], sch["sqrt_one_minus_alphas"][i], nz)
    cl = lambda t: t.permute(0, 2, 3, 1).contiguous().to(dev)
    xs, eps = cl(x), torch.zeros(2, 8, 8, 32, device=dev)
    eps[..., :4] = cl(e)
    sc = torch.tensor([float(sch["alphas"][i]), float(sch["alphas_prev"][i]), float(sch["sigmas"][i]), float(sch["sqrt_one_minus_alphas"][i])],
                      dtype=torch.float32, device=dev)
    p0 = torch.empty_like(xs)
    uin = torch.zeros(2, 8, 8, 32, dtype=torch.bfloat16, device=dev)
    ops.ddim_step(xs, eps, sc, noise=cl(nz), pred_x0_out=p0, unet_in=uin)
    assert torch.allclose(xs.cpu().permute(0, 3, 1, 2), ref, rtol=1e-6, atol=1e-6)
    assert torch.allclose(p0.cpu().permute(0, 3, 1, 2), ref0, rtol=1e-6, atol=1e-6)
    assert torch.equal(uin[..., :4].float().cpu(), xs.cpu().bfloat16().float())
    d = torch.randn(3, 1, 40, 40, generator=gen) * 3
    assert torch.allclose(ops.minmax_normalise(d.to(dev)).cpu(), S.slice_minmax_normalise(d), rtol=1e-6, atol=1e-7)


# ------------------------------------------------------------------------------------------------ blocks vs golden
def test_blocks_match_reference_fixtures(dev):
    from jointimagegeneration_amd import blocks as B
    from jointimagegeneration_amd import ops
    g = gold("modules")
    rb = seeded(B.ResBlock(64, 128, 0.0, out_channels=96, dims=3), "rb3d.").to(dev)
    tb = torch.zeros(1, 96, device=dev)
    rb.time_bias(T(g["rb3d_emb"]).to(dev), tb)
    y = rb.run(ops.to_cl(T(g["rb3d_x"]).to(dev)), tb)
    assert rel_err(ops.from_cl(y, 3), T(g["rb3d_y"])) < 3e-2
    ab = seeded(B.AttentionBlock(64, num_heads=1, num_head_channels=32), "ab3d.").to(dev)
    assert rel_err(ops.from_cl(ab.run(ops.to_cl(T(g["ab3d_x"]).to(dev))), 3), T(g["ab3d_y"])) < 3e-2
    up = seeded(B.Upsample(32, True, dims=3), "up3d.").to(dev); dn = seeded(B.Downsample(32, True, dims=3), "dn3d.").to(dev)
    x = ops.to_cl(T(g["ud3d_x"]).to(dev))
    assert rel_err(ops.from_cl(up.run(x), 3), T(g["up3d_y"])) < 2e-2
    assert rel_err(ops.from_cl(dn.run(x), 3), T(g["dn3d_y"])) < 2e-2
    rb2 = seeded(B.ResBlock(64, 128, 0.0, out_channels=64, dims=2), "rb2d.").to(dev)
    tb = torch.zeros(2, 64, device=dev)
    rb2.time_bias(T(g["rb2d_emb"]).to(dev), tb)
    assert rel_err(ops.from_cl(rb2.run(ops.to_cl(T(g["rb2d_x"]).to(dev)), tb), 2), T(g["rb2d_y"])) < 3e-2
    ab2 = seeded(B.AttentionBlock(96, num_heads=-1, num_head_channels=32), "ab2d.").to(dev)
    assert rel_err(ops.from_cl(ab2.run(ops.to_cl(T(g["ab2d_x"]).to(dev))), 2), T(g["ab2d_y"])) < 3e-2
    st = seeded(B.SpatialTransformer(64, 2, 32, depth=1, context_dim=48), "st.").to(dev)
    ctx = ops.to_cl(T(g["st_ctx"]).permute(0, 2, 1).contiguous().to(dev), c_pad=64)
    assert rel_err(ops.from_cl(st.run(ops.to_cl(T(g["st_x"]).to(dev)), ctx), 2), T(g["st_y"])) < 3e-2
    r = seeded(B.ResnetBlock(in_channels=32, out_channels=64, dropout=0.0), "aer.").to(dev)
    assert rel_err(ops.from_cl(r.run(ops.to_cl(T(g["aer_x"]).to(dev))), 2), T(g["aer_y"])) < 3e-2
    a2 = seeded(B.AttnBlock2d(64), "aea.").to(dev)
    assert rel_err(ops.from_cl(a2.run(ops.to_cl(T(g["aea_x"]).to(dev))), 2), T(g["aea_y"])) < 3e-2


@pytest.mark.parametrize("hint", [0, 1, 4, 6], ids=["production_dispatch", "halo_hint_box512", "halo_hint_box256", "halo_hint_box1024"])
def test_small_networks_match_reference_fixtures(dev, hint, monkeypatch):
    from jointimagegeneration_amd import ops
    monkeypatch.setattr(ops, "PATH_HINT", hint)
    g = gold("networks_small")
    K, u, u2, u3, ae = build_small()
    u, u2, u3, ae = u.to(dev), u2.to(dev), u3.to(dev), ae.to(dev)
    lab = T(g["ccdm_labels"]).long()
    out = u(S.one_hot_bchw(lab, K).to(dev), torch.zeros(1, 1, 8, 8, 8, device=dev), None, T(g["ccdm_t"]).to(dev))["diffusion_out"]
    assert out.shape == (1, K, 8, 8, 8)
    assert float((out.cpu() - T(g["ccdm_probs"])).abs().max()) < 1.5e-2          # probabilities: absolute tolerance (measured 6.3e-3)
    assert torch.allclose(out.sum(1).cpu(), torch.ones(1, 8, 8, 8), atol=1e-5)
    e = u2(T(g["ldm_x"]).to(dev), T(g["ldm_t"]).to(dev))
    assert rel_err(e, T(g["ldm_eps"])) < 3e-2 and rms_err(e, T(g["ldm_eps"])) < 2e-2          # measured 9.5e-3 / 9.8e-3
    e = u3(T(g["ldm_x"]).to(dev), T(g["ldm_t"]).to(dev), context=T(g["ldmst_ctx"]).to(dev))
    assert rel_err(e, T(g["ldmst_eps"])) < 4e-2 and rms_err(e, T(g["ldmst_eps"])) < 2.5e-2    # measured 1.3e-2 / 1.4e-2
    dec = ae.decode(T(g["ae_z"]).to(dev))
    assert rel_err(dec, T(g["ae_dec"])) < 4e-2 and rms_err(dec, T(g["ae_dec"])) < 2e-2        # measured 1.3e-2 / 1.1e-2
    mode = ae.encode(T(g["ae_img"]).to(dev)).mode()
    assert rel_err(mode, T(g["ae_mode"])) < 4e-2 and rms_err(mode, T(g["ae_mode"])) < 2e-2    # measured 1.7e-2 / 1.1e-2


# ------------------------------------------------------------------------------------------------ chains
def test_ccdm_chain_teacher_forced_and_graph(dev):
    """Per-step parity protocol (SURVEY 7 'hard parts' (i)): same x_t, same tape => same label except where the race is
    a near-tie under bf16 logits; the count is reported and bounded.  Then the hipGraph/Philox throughput path."""
    from jointimagegeneration_amd.ccdm import DenoisingModel, DiffusionModel
    g = gold("chains_small")
    K, u, *_ = build_small()
    model = DenoisingModel(DiffusionModel("cosine", 5, K, dims=3), u, "none", "confidence", dims=3).eval().to(dev)
    E0, tapes = T(g["ccdm_E0"]), list(T(g["ccdm_tapes"]))
    xT = S.race_sample_labels(torch.full((1, K, 8, 8, 8), 1.0 / K), E0).int()
    ref_steps = T(g["ccdm_step_labels"])                          # [5, 1, 8, 8, 8] labels after each reference step
    cond = torch.zeros(1, 1, 8, 8, 8, device=dev)
    # teacher forcing: feed the REFERENCE's x_t into every single step
    tv = model.t_values(None)
    total_mism = 0
    for i, t in enumerate(tv):
        xt = xT if i == 0 else ref_steps[i - 1].int()
        sub = DenoisingModel(DiffusionModel("cosine", 5, K, dims=3), u, "none", "confidence", dims=3).eval().to(dev)
        trace = []
        # run exactly one step starting at t: t_values(init_t=t) = [t, t-1, ...]; stop after the first via the trace
        sub.sample_labels(xt.to(dev), cond, init_t=t, rng_tapes=tapes[i:], trace=trace)
        mism = int((trace[0]["labels"].cpu() != ref_steps[i].int()).sum())
        total_mism += mism
    print(f"teacher-forced CCDM steps: {total_mism} label mismatches over {5 * 512} voxel-steps (bf16 logits vs fp32)")
    assert total_mism <= 0.02 * 5 * 512
    # free-running chain with tapes: final probabilities close to the reference's
    out = model(S.one_hot_bchw(xT.long(), K).to(dev), cond, rng_tapes=tapes)["diffusion_out"]
    agree = float((out.argmax(1).cpu() == T(g["ccdm_final_labels"]).long()).float().mean())
    print(f"free-running 5-step chain: final argmax agreement with the reference = {agree:.4f}")
    # a free-running stochastic chain amplifies ulp-level logit differences (two fp32 CPU runs of the reference and the
    # oracle already disagree on 61 % of the voxels after 50 steps of the full model: tests/golden/e2e_c1.npz), so this is
    # only a sanity bound far above chance (1/K = 0.17); the binding per-step criterion is the teacher-forced count above
    assert agree > 0.5
    # throughput path: Philox + hipGraph. Deterministic for a fixed seed, and the graph equals the eager launches.
    big = DenoisingModel(DiffusionModel("cosine", 8, K, dims=3), u, "none", "majority", dims=3).eval().to(dev)
    a, _ = big.sample_labels(xT.to(dev), cond)
    b, _ = big.sample_labels(xT.to(dev), cond)
    big.use_graph = False
    c, _ = big.sample_labels(xT.to(dev), cond)
    assert torch.equal(a, b) and torch.equal(a, c)


def test_ddim_quad_schedule_and_classifier_free_guidance_vs_reference_fixture(dev):
    """The two sampler options of DDIMSampler that no shipped config uses (VERDICT r03 missing #3), against what the REFERENCE sampler produced
    (tests/golden/ddim_options.npz, make_golden.py fx_ddim_options): the "quad" discretisation (util.py:50-52) and classifier-free guidance
    e = e_u + s (e_c - e_u) (ddim.py:175-180; two UNet evaluations per step + gg_lincomb4 here).  Tolerance: a bf16 network inside a 5 / 6
    step chain, as for the plain chain above (3 x its error at scale 3: guidance multiplies the difference of two evaluations)."""
    from jointimagegeneration_amd.ldm import DDIMSampler, LatentDiffusion
    g = gold("ddim_options")
    cfg_unet = dict(target="ldm.modules.diffusionmodules.openaimodel.UNetModel", params=dict(LDM_SMALL))
    cfg_ae = dict(target="ldm.models.autoencoder.AutoencoderKL", params=dict(embed_dim=4, dims=2, ddconfig=dict(AE_SMALL), lossconfig=dict(target="torch.nn.Identity")))
    cfg_cond = dict(target="ldm.models.autoencoder.AutoencoderKL", params=dict(embed_dim=4, dims=2, ddconfig=dict(AE_SMALL, in_channels=2, out_ch=2), lossconfig=dict(target="torch.nn.Identity")))
    m = seeded(LatentDiffusion(first_stage_config=cfg_ae, cond_stage_config=cfg_cond, unet_config=cfg_unet, linear_start=0.0015,
                               linear_end=0.0195, timesteps=1000, image_size=8, channels=4, dims=2, first_stage_key="image",
                               cond_stage_key="mask", num_timesteps_cond=1), "ldm_pipe.").to(dev)
    c, uc, x_T = T(g["c"]).to(dev), T(g["uc"]).to(dev), T(g["x_T"]).to(dev)       # the REFERENCE's conditionings: the samplers alone are compared
    sampler = DDIMSampler(m)
    z_cfg, _ = sampler.sample(S=5, batch_size=2, shape=(4, 8, 8), conditioning=c, verbose=False, x_T=x_T, dims=2,
                              unconditional_guidance_scale=3.0, unconditional_conditioning=uc)
    e1, r1 = rel_err(z_cfg, T(g["z_cfg"])), rms_err(z_cfg, T(g["z_cfg"]))
    z_plain, _ = sampler.sample(S=5, batch_size=2, shape=(4, 8, 8), conditioning=c, verbose=False, x_T=x_T, dims=2)
    assert rel_err(z_plain, T(g["z_cfg"])) > 5 * e1                                   # guidance really changed the sample
    z_one, _ = sampler.sample(S=5, batch_size=2, shape=(4, 8, 8), conditioning=c, verbose=False, x_T=x_T, dims=2,
                              unconditional_guidance_scale=1.0, unconditional_conditioning=uc)
    assert torch.equal(z_one, z_plain)                                                # scale 1 is the unguided path (ddim.py:172)
    z_quad, _ = sampler.sample(S=6, batch_size=2, shape=(4, 8, 8), conditioning=c, verbose=False, x_T=x_T, dims=2, ddim_discretize="quad")
    assert np.array_equal(sampler.ddim_timesteps, g["quad_ts_used"])
    e2, r2 = rel_err(z_quad, T(g["z_quad"])), rms_err(z_quad, T(g["z_quad"]))
    print(f"DDIM guidance scale 3: max {e1:.3e} rms {r1:.3e}; quad schedule: max {e2:.3e} rms {r2:.3e} (of the reference's max)")
    assert e1 < 6e-2 and r1 < 3e-2 and e2 < 2e-2 and r2 < 1.5e-2


def test_ldm_pipeline_ddim_chain(dev):
    from jointimagegeneration_amd.ldm import DDIMSampler, LatentDiffusion
    g = gold("chains_small")
    cfg_unet = dict(target="ldm.modules.diffusionmodules.openaimodel.UNetModel", params=dict(LDM_SMALL))
    cfg_ae = dict(target="ldm.models.autoencoder.AutoencoderKL", params=dict(embed_dim=4, dims=2, ddconfig=dict(AE_SMALL), lossconfig=dict(target="torch.nn.Identity")))
    cfg_cond = dict(target="ldm.models.autoencoder.AutoencoderKL", params=dict(embed_dim=4, dims=2, ddconfig=dict(AE_SMALL, in_channels=2, out_ch=2), lossconfig=dict(target="torch.nn.Identity")))
    m = seeded(LatentDiffusion(first_stage_config=cfg_ae, cond_stage_config=cfg_cond, unet_config=cfg_unet, linear_start=0.0015,
                               linear_end=0.0195, timesteps=1000, image_size=8, channels=4, dims=2, first_stage_key="image",
                               cond_stage_key="mask", num_timesteps_cond=1), "ldm_pipe.").to(dev)
    c = m.get_learned_conditioning(T(g["ldm_concat_cond"]).to(dev))
    assert rel_err(c, T(g["ldm_c"])) < 3e-2                                          # measured 1.3e-2
    sampler = DDIMSampler(m)
    z, _ = sampler.sample(S=5, batch_size=2, shape=(4, 8, 8), conditioning=c, verbose=False, x_T=T(g["ldm_x_T"]).to(dev), dims=2,
                          noise_tape=list(T(g["ldm_noises"])))
    assert rel_err(z, T(g["ldm_z"])) < 2e-2 and rms_err(z, T(g["ldm_z"])) < 1.5e-2           # measured 6.6e-3 / 5.4e-3
    dec = m.decode_first_stage(z)
    assert rel_err(dec, T(g["ldm_dec"])) < 3e-2 and rms_err(dec, T(g["ldm_dec"])) < 2.5e-2     # measured 9.6e-3 / 1.2e-2
    # PLMS sampler on the same engine (plms.py:118-236): Euler start + Adams-Bashforth 2..4
    from jointimagegeneration_amd.ldm import PLMSSampler
    zp, _ = PLMSSampler(m).sample(S=10, batch_size=2, shape=(4, 8, 8), conditioning=c, verbose=False, x_T=T(g["ldm_x_T"]).to(dev))
    assert rel_err(zp, T(g["ldm_plms_z"])) < 1.5e-2 and rms_err(zp, T(g["ldm_plms_z"])) < 1e-2   # measured 4.3e-3 / 3.6e-3
    with pytest.raises(ValueError, match="must be 0 for PLMS"):
        PLMSSampler(m).sample(S=10, batch_size=2, shape=(4, 8, 8), conditioning=c, eta=0.5)
    # vanilla ancestral sampling (LatentDiffusion.p_sample_loop) on a 20-step schedule, same networks (same parameter names)
    m20 = seeded(LatentDiffusion(first_stage_config=cfg_ae, cond_stage_config=cfg_cond, unet_config=cfg_unet, linear_start=0.0015,
                                 linear_end=0.0195, timesteps=20, image_size=8, channels=4, dims=2, first_stage_key="image",
                                 cond_stage_key="mask", num_timesteps_cond=1), "ldm_pipe.").to(dev)
    zv = m20.p_sample_loop(c, (2, 4, 8, 8), x_T=T(g["ldm_x_T"]).to(dev), verbose=False, noise_tape=list(T(g["ldm_vanilla_noises"])))
    assert rel_err(zv, T(g["ldm_vanilla_z"])) < 1.5e-2 and rms_err(zv, T(g["ldm_vanilla_z"])) < 1e-2   # measured 3.0e-3 / 3.6e-3
    # hipGraph path == eager path, bit for bit (eta = 0, no tape)
    s2 = DDIMSampler(m)
    za, _ = s2.sample(S=5, batch_size=2, shape=(4, 8, 8), conditioning=c, verbose=False, x_T=T(g["ldm_x_T"]).to(dev), dims=2)
    s3 = DDIMSampler(m); s3.use_graph = False
    zb, _ = s3.sample(S=5, batch_size=2, shape=(4, 8, 8), conditioning=c, verbose=False, x_T=T(g["ldm_x_T"]).to(dev), dims=2)
    assert torch.equal(za, zb)
    zc, _ = s2.sample(S=5, batch_size=2, shape=(4, 8, 8), conditioning=c, verbose=False, x_T=T(g["ldm_x_T"]).to(dev), dims=2)
    assert torch.equal(za, zc)          # first call eager, second call = one captured graph of the whole chain
    assert next(iter(s2._graphs.values()))["graph"] is not None
    zc2, _ = s2.sample(S=5, batch_size=2, shape=(4, 8, 8), conditioning=c, verbose=False, x_T=T(g["ldm_x_T"]).to(dev), dims=2)
    assert torch.equal(za, zc2)         # replaying the cached graph on fresh inputs
    # DDIM update as the head conv's epilogue == the separate gg_ddim_step launch, bit for bit (latent AND pred_x0)
    s4 = DDIMSampler(m); s4.fuse_ddim = False
    zd, inter_d = s4.sample(S=5, batch_size=2, shape=(4, 8, 8), conditioning=c, verbose=False, x_T=T(g["ldm_x_T"]).to(dev), dims=2)
    s5 = DDIMSampler(m)
    ze, inter_e = s5.sample(S=5, batch_size=2, shape=(4, 8, 8), conditioning=c, verbose=False, x_T=T(g["ldm_x_T"]).to(dev), dims=2)
    assert s4.last_step_fused is False and torch.equal(zd, ze) and torch.equal(inter_d["pred_x0"][1], inter_e["pred_x0"][1])
    assert torch.equal(zd, za)


# ------------------------------------------------------------------------------------------------ glue + entry points
def test_mask_to_cond_slice_matches_scipy_zoom_recipe(dev):
    """Stage glue, integer path => bit-exact vs the reference recipe rot90(scipy.ndimage.zoom(mask, target/shape, order=0), k=3)/255
    (latentdiffusion/sample_diffusion.py:199-200): (a) the committed fixture made by scipy in the build container (non-integer
    ratios on every axis), (b) scipy itself run on the box, (c) the oracle restatement of the index rule."""
    from scipy.ndimage import zoom
    from jointimagegeneration_amd import ops
    from util import synth_labels
    g = gold("glue")
    lab = torch.from_numpy(synth_labels((10, 12, 14), 12, seed=3)).int()
    D, H, W = 23, 32, 32
    want = T(g["small_rot_labels"]).float() / 255.0                                       # [D, H, W], made by scipy.zoom + rot90
    live = torch.rot90(torch.from_numpy(zoom(lab.numpy(), np.array((D, H, W)) / np.array(lab.shape), order=0)), dims=(1, 2), k=3).float() / 255.0
    assert torch.equal(want, live) and torch.equal(want, S.mask_to_cond_volume(lab.long(), (D, H, W)))
    lab2 = torch.stack([lab, torch.flip(lab, dims=(0, 1))])                                # batch of 2 different volumes
    want2 = torch.stack([want, S.mask_to_cond_volume(lab2[1].long(), (D, H, W))])
    gen = torch.Generator().manual_seed(21)
    prev = torch.rand(2, H, W, generator=gen)
    cond = torch.empty(2, 1, H, W, 32, dtype=torch.bfloat16, device=dev)
    mo = torch.empty(2, H, W, device=dev)
    for m in range(D):
        ops.mask_to_cond_slice(lab2.to(dev), m, D, H, W, prev.to(dev), cond, mask_out=mo)
        assert torch.equal(mo.cpu(), want2[:, m]), m
        assert torch.equal(cond[:, 0, :, :, 1].float().cpu(), want2[:, m].bfloat16().float())
        assert torch.equal(cond[:, 0, :, :, 0].float().cpu(), prev.bfloat16().float())
        assert float(cond[..., 2:].float().abs().max()) == 0.0
    # the rule differs from F.interpolate(nearest) (which round 1 had implemented): make sure the test can tell them apart
    up = F.interpolate(lab[None, None].float(), (D, H, W), mode="nearest")[0, 0]
    assert not torch.equal(torch.rot90(up, k=3, dims=(1, 2)) / 255.0, want)


def test_mask_to_cond_slice_full_size(dev):
    """BASELINE size: 128^3 CCDM labels -> 256 slices of 512x512; every voxel against scipy.ndimage.zoom run on the box, and the
    per-slice position-weighted checksums against the fixture written in the build container."""
    from scipy.ndimage import zoom
    from jointimagegeneration_amd import ops
    from util import synth_labels
    g = gold("glue")
    lab_np = synth_labels((128, 128, 128), 12, seed=7)
    D, H, W = 256, 512, 512
    live = torch.rot90(torch.from_numpy(zoom(lab_np, np.array((D, H, W)) / np.array(lab_np.shape), order=0)), dims=(1, 2), k=3)
    lab = torch.from_numpy(lab_np).int()[None].to(dev)
    cond = torch.empty(1, 1, H, W, 32, dtype=torch.bfloat16, device=dev)
    vol = torch.empty(D, H, W, device=dev)
    for m in range(D):
        ops.mask_to_cond_slice(lab, m, D, H, W, None, cond, mask_out=vol[m:m + 1])
    got = torch.round(vol * 255.0).long().cpu()
    assert torch.equal(vol.cpu(), live.float() / 255.0)
    i = torch.arange(512)[:, None]; j = torch.arange(512)[None, :]
    wgt = ((i * 7 + j * 13) % 31 + 1).long()
    assert torch.equal(got.sum((1, 2)), T(g["full_slice_sum"])) and torch.equal((got * wgt[None]).sum((1, 2)), T(g["full_slice_wsum"]))


def test_entry_points_run_on_small_configs(dev, tmp_path):
    """ddpm_eval / sample_diffusion keep the reference's CLI + yaml schema (+ dotted `target:` paths) end to end."""
    import yaml
    from jointimagegeneration_amd import ddpm_eval, sample_diffusion
    params = dict(output_path=str(tmp_path), exp_name="t", evaluation_vote_strategy="confidence", dataset_file="datasets.ruijin",
                  batch_size=3, dims=3, beta_schedule="cosine", beta_schedule_params=dict(s=0.008), time_steps=6,
                  backbone="unet_openai", feature_cond_encoder=dict(type="none"),
                  unet_openai=dict(base_channels=32, channel_mult=[1, 2, 2], attention_resolutions=[2, 4], num_heads=1,
                                   num_head_channels=32, softmax_output=True),
                  load_from="/mnt/does/not/exist.pt")
    pf = tmp_path / "params_eval.yml"
    pf.write_text(yaml.safe_dump(params))
    ddpm_eval.main([str(pf), "exp", "--size", "8", "8", "8", "--num-classes", "6", "--num-volumes", "2"])
    outs = sorted((tmp_path / "exp").glob("pred_*.nii.gz"))
    assert len(outs) == 2
    ae = lambda cin: dict(target="ldm.models.autoencoder.AutoencoderKL",
                          params=dict(ckpt_path="/mnt/none/last.ckpt", embed_dim=4, monitor="val/rec_loss", dims=2,
                                      ddconfig=dict(AE_SMALL, in_channels=cin, out_ch=cin), lossconfig=dict(target="torch.nn.Identity")))
    cfg = dict(model=dict(base_learning_rate=2e-6, target="ldm.models.diffusion.ddpm.LatentDiffusion",
                          params=dict(linear_start=0.0015, linear_end=0.0195, num_timesteps_cond=1, log_every_t=200, timesteps=1000,
                                      first_stage_key="image", cond_stage_key="mask", image_size=8, channels=4, dims=2,
                                      monitor="val/loss_simple_ema",
                                      unet_config=dict(target="ldm.modules.diffusionmodules.openaimodel.UNetModel", params=dict(LDM_SMALL)),
                                      first_stage_config=ae(1), cond_stage_config=ae(2))))
    logdir = tmp_path / "logs" / "run"
    (logdir / "configs").mkdir(parents=True)
    (logdir / "configs" / "project.yaml").write_text(yaml.safe_dump(cfg))
    sample_diffusion.main(["-r", str(logdir), "-c", "5", "-n", "2", "--slices", "4", "--size", "32"])
    outs = sorted((logdir / "samples" / "00000000").glob("sample_*.nii.gz"))
    assert len(outs) == 2


# ------------------------------------------------------------------------------------------------ BASELINE full sizes
def test_full_size_conv3d_properties(dev):
    """128^3 x 64 -> 64 conv (the dominant launch of config C3/C5) is too large for the CPU oracle in a unit test, so it is
    checked through size-independent properties that are EXACT in bf16: homogeneity under power-of-two scaling, shift
    equivariance away from the border (exercises every halo/tile seam), repeatability, and agreement with the oracle on a
    cropped sub-volume whose receptive field lies inside the crop."""
    from jointimagegeneration_amd import ops
    g = torch.Generator(device=dev).manual_seed(3)
    S = 128
    x = torch.randn(1, S, S, S, 64, generator=g, device=dev).bfloat16()
    w = (torch.randn(64, 64, 3, 3, 3, generator=g, device=dev) / math.sqrt(64 * 27))
    pw = ops.pack_conv_weight(w, 64)
    assert ops.conv_runs_halo_tile(ops.CL(x, 64), 64, k=(3, 3, 3))            # this shape runs on the halo-tile kernel
    y = ops.conv(ops.CL(x, 64), pw, None, 64, k=(3, 3, 3), out_f32=True).t
    y2 = ops.conv(ops.CL(x, 64), pw, None, 64, k=(3, 3, 3), out_f32=True).t
    assert torch.equal(y, y2)                                                   # repeatable bit for bit
    ys = ops.conv(ops.CL(x * 4, 64), pw, None, 64, k=(3, 3, 3), out_f32=True).t
    assert torch.equal(ys, y * 4)                                               # exact: x4 only changes exponents
    sh = (5, 9, 17)                                                             # crosses tile seams in D, H and W
    xr = torch.roll(x, shifts=sh, dims=(1, 2, 3))
    yr = ops.conv(ops.CL(xr, 64), pw, None, 64, k=(3, 3, 3), out_f32=True).t
    a = yr[:, sh[0] + 1:S - 1, sh[1] + 1:S - 1, sh[2] + 1:S - 1]
    b = y[:, 1:S - 1 - sh[0], 1:S - 1 - sh[1], 1:S - 1 - sh[2]]
    assert torch.equal(a, b)                                                    # shift equivariance in the interior
    # oracle on a crop: outputs [40:56, 60:76, 100:116] only see inputs [39:57, 59:77, 99:117]
    xc = x[:, 39:57, 59:77, 99:117].float().permute(0, 4, 1, 2, 3).cpu()
    ref = O.conv(xc, bf(w.cpu()), None, padding=0)
    got = y[:, 40:56, 60:76, 100:116, :64].permute(0, 4, 1, 2, 3).cpu()
    assert rel_err(got, ref) < 1e-2


def test_full_size_posterior_properties(dev):
    """2 097 152 voxels, K = 14 (config C3): at t = 1 the fused kernel returns argmax of the UNet probabilities; labels are
    in range; the Philox race is reproducible and its label histogram follows the mean posterior."""
    from jointimagegeneration_amd import ops
    K, M = 14, 128 ** 3
    g = torch.Generator(device=dev).manual_seed(5)
    logits = 2 * torch.randn(M, 32, generator=g, device=dev)
    lab = torch.randint(0, K, (M,), generator=g, device=dev, dtype=torch.int32)
    one = torch.tensor([0.0, 1.0], device=dev)                                  # t == 1: a = 0, abar = 1
    out = ops.ccdm_posterior_sample(logits, True, lab, one, K, draw=False)
    assert torch.equal(out.long(), logits[:, :K].argmax(-1))
    sc = torch.tensor([0.97, 0.6], device=dev)
    off = torch.tensor([123], dtype=torch.int64, device=dev)
    probs = torch.empty(M, K, device=dev)
    a = ops.ccdm_posterior_sample(logits, True, lab, sc, K, philox_seed=9, philox_offset=off, draw=True, probs_out=probs)
    b = ops.ccdm_posterior_sample(logits, True, lab, sc, K, philox_seed=9, philox_offset=off, draw=True)
    assert torch.equal(a, b) and int(a.min()) >= 0 and int(a.max()) < K
    assert torch.allclose(probs.sum(-1), torch.ones(M, device=dev), atol=1e-5)
    freq = torch.bincount(a.long(), minlength=K).float() / M
    assert float((freq - probs.mean(0)).abs().max()) < 2e-3


# ------------------------------------------------------------------------------------------------ fp32 validation kernels
def test_fp32_validation_kernels_vs_torch(dev):
    """gg_f32.hip (fp32 validation mode of the CCDM path) against ATen fp32 on the CPU: 3-D convs (3x3x3 stride 1 / 2, fused x2
    upsample, 1x1x1, skip concat as second source, per-sample bias, residual, channel padding), GroupNorm(+SiLU) with one and two
    sources, QKVAttentionLegacy.  Tolerance: a few fp32 ulps of the output scale (different summation orders only)."""
    import torch.nn.functional as F
    from jointimagegeneration_amd import ops
    g = torch.Generator().manual_seed(4242)
    tol = 3e-6
    with ops.fp32_validation():
        x = torch.randn(2, 15, 6, 8, 10, generator=g)
        x2 = torch.randn(2, 64, 6, 8, 10, generator=g)
        xcl, x2cl = ops.to_cl(x.to(dev)), ops.to_cl(x2.to(dev))
        assert xcl.t.dtype == torch.float32 and xcl.Cpad == 32
        for (cout, k, stride, up) in ((40, 3, 1, False), (64, 3, 2, False), (33, 3, 1, True), (96, 1, 1, False)):
            w = torch.randn(cout, 15, k, k, k, generator=g) / math.sqrt(15 * k ** 3)
            b = torch.randn(2, cout, generator=g)
            xin = F.interpolate(x, scale_factor=2, mode="nearest") if up else x
            ref = F.conv3d(xin, w, None, stride=stride, padding=k // 2) + b[:, :, None, None, None]
            bias = torch.zeros(2, ops.pad32(cout)); bias[:, :cout] = b
            got = ops.conv(xcl, ops.pack_conv_weight(w.to(dev), 32), bias.to(dev), cout, k=(k, k, k), stride=stride, pad=k // 2, upsample=up,
                           bias_per_sample=True)
            assert got.t.dtype == torch.float32 and float(got.t[..., cout:].abs().max() if got.Cpad > cout else 0.0) == 0.0
            assert rel_err(ops.from_cl(got, 3), ref) < tol, (cout, k, stride, up)
        # skip concat (second source) + residual
        w = torch.randn(64, 64 + 64, 3, 3, 3, generator=g) / math.sqrt(128 * 27)
        res = torch.randn(2, 64, 6, 8, 10, generator=g)
        x1 = torch.randn(2, 64, 6, 8, 10, generator=g)
        ref = F.conv3d(torch.cat([x1, x2], 1), w, None, padding=1) + res
        got = ops.conv(ops.to_cl(x1.to(dev)), ops.pack_conv_weight(w.to(dev), 128), None, 64, k=(3, 3, 3), src2=x2cl, residual=ops.to_cl(res.to(dev)))
        assert rel_err(ops.from_cl(got, 3), ref) < tol
        # GroupNorm (+SiLU), one source with an offset mean, and the two-source concat (groups straddle the boundary: 96 + 64 = 160 / 32 = 5)
        xa = torch.randn(2, 96, 4, 6, 6, generator=g) * 1.5 + 3.0
        xb = torch.randn(2, 64, 4, 6, 6, generator=g)
        gam, bet = 1 + 0.1 * torch.randn(160, generator=g), 0.1 * torch.randn(160, generator=g)
        for act in (True, False):
            ref = F.group_norm(xa, 32, gam[:96], bet[:96], 1e-5)
            ref = F.silu(ref) if act else ref
            got = ops.groupnorm_f32(ops.to_cl(xa.to(dev)), gam[:96].contiguous().to(dev), bet[:96].contiguous().to(dev), 1e-5, act)
            assert rel_err(ops.from_cl(got, 3), ref) < tol
            ref = F.group_norm(torch.cat([xa, xb], 1), 32, gam, bet, 1e-5)
            ref = F.silu(ref) if act else ref
            got = ops.groupnorm_f32(ops.to_cl(xa.to(dev)), gam.to(dev), bet.to(dev), 1e-5, act, src2=ops.to_cl(xb.to(dev)))
            assert rel_err(ops.from_cl(got, 3), ref) < tol
        # AttentionBlock (legacy qkv order) as a whole module, fp32
        from jointimagegeneration_amd.blocks import AttentionBlock
        ab = seeded(AttentionBlock(64, num_heads=1, num_head_channels=32), "ab3d.")
        gm = gold("modules")
        ref = T(gm["ab3d_y"])
        got = ab.to(dev).run(ops.to_cl(T(gm["ab3d_x"]).to(dev)))
        assert got.t.dtype == torch.float32
        e = rel_err(ops.from_cl(got, 3), ref)
        print(f"fp32 validation AttentionBlock vs the reference fixture: {e:.2e}")
        assert e < 1e-5
    # leaving the mode: the production (bf16) packs are used again
    assert ops.to_cl(x.to(dev)).t.dtype == torch.bfloat16


@pytest.mark.parametrize("cfg", [(1, 640, 1920, (16, 16), 1, False), (1, 640, 1920, (8, 8), 1, False), (2, 320, 960, (8, 16), 1, False),
                                 (1, 160, 4, (64, 64), 3, True), (1, 1280, 640, (8, 8), 1, False)],
                         ids=["qkv_16x16", "qkv_8x8", "qkv_n2", "head_64x64_silu", "two_stage_1x1"])
def test_conv_groupnorm_prologue_from_accumulators(dev, cfg, monkeypatch):
    """gg_conv_desc.pro_acc1: a box-kernel conv folds the (sum, sumsq) accumulators its producer left into the GroupNorm scale / shift
    table itself and normalises (* SiLU) its staged box in place -- no GroupNorm launch.  Producer = a real box conv emitting the sums;
    compared with the oracle's conv(act(GroupNorm(y))) on the stored bf16 tensor, with the launch-based path, and twice for
    bit-reproducibility; two-source concat for the cases that allow it."""
    from jointimagegeneration_amd import ops
    N, C, Cout, sp, k, act = cfg
    g = torch.Generator().manual_seed(C + Cout + k)
    x = torch.randn((N, C) + sp, generator=g)
    w0 = torch.randn(C, C, 1, 1, generator=g) / math.sqrt(C)
    w = torch.randn(Cout, C, k, k, generator=g) / math.sqrt(C * k * k)
    b = torch.randn(Cout, generator=g) * 0.1
    gamma, beta = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    xcl = ops.to_cl(x.to(dev))
    ops.stats_begin(dev)
    try:
        y = ops.conv(xcl, ops.pack_conv_weight(w0.to(dev), xcl.Cpad), None, C, k=(1, 1, 1), pad=0, want_stats=True)      # producer (box 1x1 / gather5)
        if y.acc is None or y.acc.shape[1] != 1:
            pytest.skip("producer shape does not emit 1-stripe sums")
        kw = dict(k=(1, k, k), pad=k // 2)
        assert ops.conv_prologue_from_acc(y, Cout, act, **kw)
        # SiLU norms are only folded where at most two cout tiles share a box (the head conv): a wide SiLU conv must say no
        assert not ops.conv_prologue_from_acc(y, 640, True, k=(1, 3, 3), pad=1) or C * sp[0] * sp[1] == 0
        pw, pb = ops.pack_conv_weight(w.to(dev), y.Cpad), ops.pad_bias(b.to(dev), Cout, dev)
        outs = [ops.conv(y, pw, pb, Cout, prologue_acc=(gamma.to(dev), beta.to(dev), 1e-5), prologue_silu=act, **kw) for _ in range(2)]
        assert torch.equal(outs[0].t, outs[1].t)
        yf = ops.from_cl(y, 2).cpu()
        a = O.group_norm(yf, gamma, beta, 1e-5)
        a = O.silu(a) if act else a
        ref = O.conv(bf(a), bf(w), b, padding=k // 2)
        got = ops.from_cl(outs[0], 2).cpu()
        assert rel_err(got, ref) < 1.5e-2, rel_err(got, ref)
        # launch-based path on the same tensor: statistics pass + apply + conv
        sc, sh = ops.groupnorm_stats(y, gamma.to(dev), beta.to(dev), 1e-5)
        old = ops.conv(ops.groupnorm_apply(y, sc, sh, act), pw, pb, Cout, **kw)
        assert rel_err(got, ops.from_cl(old, 2).cpu()) < 1e-2
    finally:
        ops.stats_end(dev)


@pytest.mark.parametrize("cfg", [(1, 160, 160, 160, (64, 64)), (1, 640, 640, 640, (16, 16)), (2, 320, 640, 0, (8, 8)), (1, 800, 800, 800, (4, 4)),
                                 (1, 320, 160, 160, (32, 32)), (1, 64, 96, 0, (20, 16))],
                         ids=["out64_160+160", "out16_640+640_two_stage_skip", "in8_320_n2", "out4_800+800", "out32_160+160", "ragged_20x16"])
def test_conv_box_kconcat_skip_projection(dev, cfg):
    """gg_conv_desc.skip_src1: out = conv3x3(a) + conv1x1(cat[x1, x2]) + (b + bs) in ONE box-kernel launch (ResBlock conv2 with its
    skip_connection, unet.py:228-262 / openaimodel.py:244-278), vs the oracle and vs the two-launch form (skip conv, then conv2 with the
    residual), one and two skip sources, several LDS stages of the x tile, a ragged last row tile."""
    from jointimagegeneration_amd import ops
    N, Cout, C1, C2, sp = cfg
    g = torch.Generator().manual_seed(Cout + C1 + C2 + sp[0])
    a = torch.randn((N, Cout) + sp, generator=g)
    x1 = torch.randn((N, C1) + sp, generator=g)
    x2 = torch.randn((N, C2) + sp, generator=g) if C2 else None
    w = torch.randn(Cout, Cout, 3, 3, generator=g) / math.sqrt(Cout * 9)
    wsk = torch.randn(Cout, C1 + C2, 1, 1, generator=g) / math.sqrt(C1 + C2)
    b, bs = 0.1 * torch.randn(Cout, generator=g), 0.1 * torch.randn(Cout, generator=g)
    acl, x1cl = ops.to_cl(a.to(dev)), ops.to_cl(x1.to(dev))
    x2cl = ops.to_cl(x2.to(dev)) if C2 else None
    assert ops.conv_fuses_skip(acl, Cout, x1cl, x2cl, k=(1, 3, 3))
    pw = ops.pack_conv_weight(w.to(dev), acl.Cpad)
    pws = ops.pack_conv_weight(wsk.to(dev), x1cl.Cpad + (x2cl.Cpad if C2 else 0))
    xin = torch.cat([x1, x2], 1) if C2 else x1
    ref = O.conv(bf(a), bf(w), b, padding=1) + O.conv(bf(xin), bf(wsk), bs)
    got = ops.conv(acl, pw, ops.pad_bias((b + bs).to(dev), Cout, dev), Cout, k=(1, 3, 3), skip=(x1cl, x2cl, pws))
    assert rel_err(ops.from_cl(got, 2), ref) < 1.2e-2, rel_err(ops.from_cl(got, 2), ref)
    res = ops.conv(x1cl, pws, ops.pad_bias(bs.to(dev), Cout, dev), Cout, k=(1, 1, 1), pad=0, src2=x2cl)
    two = ops.conv(acl, pw, ops.pad_bias(b.to(dev), Cout, dev), Cout, k=(1, 3, 3), residual=res)
    # the two-launch form rounds the skip projection to bf16 before adding it: the fused form is the more accurate one
    assert rel_err(ops.from_cl(got, 2), ops.from_cl(two, 2).cpu()) < 1.2e-2


def test_dynamic_lds_attribute_is_set_per_device(dev):
    """the MaxDynamicSharedMemorySize attribute of the 1024-position halo kernel (124 KiB) and of the box kernel is set per
    device ordinal, not once per process: a conv on a second visible device must launch.  Skipped on one-GPU boxes."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible devices")
    from jointimagegeneration_amd import ops
    g = torch.Generator().manual_seed(5)
    outs = []
    for d in (torch.device("cuda:0"), torch.device("cuda:1")):
        with torch.cuda.device(d):
            x = ops.to_cl(torch.randn(1, 32, 8, 8, 16, generator=torch.Generator().manual_seed(5)).to(d))
            w = torch.randn(32, 32, 3, 3, 3, generator=torch.Generator().manual_seed(6)).to(d) / 30.0
            import jointimagegeneration_amd.ops as O2
            old = O2.PATH_HINT
            O2.PATH_HINT = 6                                   # the 1024-position box (dynamic LDS above the default limit)
            try:
                y = ops.conv(x, ops.pack_conv_weight(w, 32), None, 32, k=(3, 3, 3))
            finally:
                O2.PATH_HINT = old
            torch.cuda.synchronize(d)
            outs.append(y.t.float().cpu())
    assert torch.equal(outs[0], outs[1])


def test_invalidate_caches_after_data_writes(dev):
    """writes through `p.data` (as the reference's ema.py does) bypass the version counter that keys the repacked-weight
    caches; `ops.invalidate_caches(module)` makes the next forward repack."""
    from jointimagegeneration_amd import ops
    from jointimagegeneration_amd.unet import UNetModel
    u = seeded(UNetModel(**LDM_SMALL), "ldm_small.").to(dev)
    g = torch.Generator().manual_seed(1)
    x, t = torch.randn(1, 8, 16, 16, generator=g).to(dev), torch.tensor([481], device=dev)
    y0 = u(x, t)
    with torch.no_grad():
        u.out[2].weight.data.mul_(2.0)                          # invisible to p._version
        u.out[2].bias.data.mul_(2.0)
    ops.invalidate_caches(u)
    y1 = u(x, t)
    assert float((y1 - 2.0 * y0).abs().max()) <= 2e-2 * float(y1.abs().max())      # head conv is linear in its weights


@pytest.mark.gpu
@pytest.mark.parametrize("case", [
    # N, C1, C2, Cout, spatial, prologue, silu, residual, per-sample bias
    (1, 64, 0, 64, (8, 8, 16), False, True, False, False),          # one item, two chunks, no prologue
    (2, 64, 32, 128, (8, 16, 32), True, False, True, True),         # two sources, two cout groups, N = 2, affine-only prologue, residual
    (1, 128, 64, 64, (24, 8, 48), True, True, False, True),         # six chunks, odd tile counts, GroupNorm * SiLU prologue
    (1, 96, 0, 192, (32, 64, 64), True, True, True, False),         # 384 items on 256 workgroups: ragged persistent loop
], ids=["one_item", "two_sources_n2", "six_chunks", "ragged_persistent"])
def test_team_halo_conv_bit_identical_to_halo_kernel(dev, case):
    """The team kernel (gg_conv_halo3.hip, path_hint 7: hand-scheduled tap phase, LDS-DMA staging, antiphase teams) against
    conv_halo_kernel (path_hint 1) on the same inputs: same MFMA, same tap / chunk order per accumulator, same epilogue arithmetic =>
    bit-identical outputs; the GroupNorm sums it leaves are the same exact integers where both kernels tile a wave alike, equal to
    fp32 rounding of the partial sums otherwise."""
    from jointimagegeneration_amd import ops
    N, C1, C2, Cout, sp, pro, silu, res, per_sample = case
    g = torch.Generator().manual_seed(zlib.crc32(repr(case).encode()) % 1000)
    x1 = ops.CL(torch.randn((N,) + sp + (C1,), generator=g).to(dev).bfloat16(), C1)
    x2 = ops.CL(torch.randn((N,) + sp + (C2,), generator=g).to(dev).bfloat16(), C2) if C2 else None
    w = torch.randn(Cout, C1 + C2, 3, 3, 3, generator=g).to(dev) / math.sqrt((C1 + C2) * 27)
    pw = ops.pack_conv_weight(w, C1 + C2)
    cp = ops.pad32(Cout)
    if per_sample:
        bias = torch.zeros(N, cp, device=dev); bias[:, :Cout] = torch.randn(N, Cout, generator=g).to(dev)
    else:
        bias = ops.pad_bias(torch.randn(Cout, generator=g).to(dev), Cout, dev)
    r = ops.CL(torch.randn((N,) + sp + (cp,), generator=g).to(dev).bfloat16(), Cout) if res else None
    prol = None
    if pro:
        gamma, beta = (1 + 0.1 * torch.randn(C1 + C2, generator=g)).to(dev), (0.1 * torch.randn(C1 + C2, generator=g)).to(dev)
        prol = ops.groupnorm_stats(x1, gamma, beta, 1e-5, src2=x2)
    outs, sums = {}, {}
    old = ops.PATH_HINT
    try:
        for hint in (1, 7):
            ops.PATH_HINT = hint
            ops.stats_begin(dev)
            y = ops.conv(x1, pw, bias, Cout, k=(3, 3, 3), src2=x2, residual=r, bias_per_sample=per_sample, prologue=prol, prologue_silu=silu)
            sums[hint] = y.acc.sum(1).clone()
            ops.stats_end(dev)
            outs[hint] = y.t.clone()
    finally:
        ops.PATH_HINT = old
    assert torch.equal(outs[1], outs[7])
    assert float((sums[1] - sums[7]).abs().max()) <= 2e-6 * float(sums[1].abs().max())


@pytest.mark.gpu
def test_stats_arena_is_replay_safe_when_a_forward_is_captured_cold(dev):
    """ops.stats_begin zeroed only the arena prefix below the high-water mark, so a forward CAPTURED without an eager
    run before it (mark 0: no memset captured at all) replayed onto its own previous GroupNorm sums.  While capturing, the whole arena is
    zeroed now: a conv that leaves its sums (gg_conv_desc.gn_acc), captured cold and replayed three times, must leave the same sums as
    the eager launch every time."""
    from jointimagegeneration_amd import ops
    g = torch.Generator().manual_seed(31)
    x = ops.CL(torch.randn((1, 1, 32, 32, 160), generator=g).to(dev).bfloat16(), 160)
    w = torch.randn(160, 160, 3, 3, generator=g).to(dev) / math.sqrt(160 * 9)
    pw = ops.pack_conv_weight(w, 160)
    pb = ops.pad_bias(None, 160, dev)
    keep = {}

    def fwd():
        ops.stats_begin(dev)
        y = ops.conv(x, pw, pb, 160, k=(1, 3, 3), want_stats=True)
        ops.stats_end(dev)
        keep["y"] = y
        return y

    ops._ARENAS.pop(str(dev), None)                  # a process that has never run a forward: high-water mark 0
    ops.stats_begin(dev); ops.stats_end(dev)         # (only the arena allocation, outside the capture)
    assert ops._ARENAS[str(dev)]["hi"] == 0
    torch.cuda.synchronize()
    graph = ops.capture_graph(fwd)                   # cold capture: no eager forward before it
    acc_view = keep["y"].acc
    assert acc_view is not None, "the box conv did not leave its sums"
    sums = []
    for _ in range(3):
        graph.replay()
        torch.cuda.synchronize()
        sums.append(acc_view.clone())
    torch.cuda.synchronize()
    y = fwd()
    torch.cuda.synchronize()
    want = y.acc.clone()
    assert int(want.abs().sum()) != 0
    for i, s_ in enumerate(sums):
        assert torch.equal(s_, want), f"replay {i}: GroupNorm sums differ from the eager launch (stale accumulators folded in)"


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["philox", "tape", "argmax"])
def test_head_conv_with_fused_ccdm_reverse_step_equals_conv_then_sampler_kernel(dev, monkeypatch, mode):
    """gg_conv_desc.post_xt: softmax + posterior + draw as the epilogue of the UNet head conv (1024-position 3-D halo box) must give,
    bit for bit, the labels and one-hot rows of the two-launch form (fp32 logits, then gg_ccdm_posterior_sample): both run
    gg_posterior.h on the same fp32 accumulator + bias.  Reference semantics: ccdm/ddpm/models/diffusion_model (posterior), pinned against
    the oracle by the sampler-kernel tests above."""
    from jointimagegeneration_amd import ops
    monkeypatch.setattr(ops, "PATH_HINT", 6)                 # the 1024-position box on a small grid
    K, Cin, sp = 14, 64, (16, 16, 32)
    M = sp[0] * sp[1] * sp[2]
    g = torch.Generator().manual_seed(77)
    x = ops.CL(torch.randn((1,) + sp + (Cin,), generator=g).to(dev).bfloat16(), Cin)
    w = (torch.randn(K, Cin, 3, 3, 3, generator=g) * (3.0 / math.sqrt(Cin * 27))).to(dev)      # logits spread over a few units
    pw = ops.pack_conv_weight(w, Cin)
    bias = torch.zeros(1, 32, device=dev); bias[0, :K] = torch.randn(K, generator=g).to(dev) * 0.3
    gamma, beta = (1 + 0.1 * torch.randn(Cin, generator=g)).to(dev), (0.1 * torch.randn(Cin, generator=g)).to(dev)
    prol = ops.groupnorm_stats(x, gamma, beta, 1e-5)
    xt = torch.randint(0, K, (M,), generator=g, dtype=torch.int32).to(dev)
    scal = torch.tensor([0.97, 0.41], dtype=torch.float32, device=dev)
    E = (-torch.log(torch.rand(M, K, generator=g).clamp_min(1e-7))).to(dev) if mode == "tape" else None
    off = torch.tensor([123], dtype=torch.int64, device=dev)
    draw = mode != "argmax"
    kw = dict(k=(3, 3, 3), out_f32=True, prologue=prol, bias_per_sample=True)

    logits = ops.conv(x, pw, bias, K, **kw).t
    lab_a = torch.empty(M, dtype=torch.int32, device=dev)
    oh_a = torch.full((M, 32), 7.0, dtype=torch.bfloat16, device=dev)
    ops.ccdm_posterior_sample(logits.view(M, -1), True, xt, scal, K, E=E, philox_seed=991, philox_offset=off, draw=draw, labels_out=lab_a, onehot_out=oh_a)

    lab_b = xt.clone()                                        # in place, as the sampler loop uses it (labels_out aliases xt)
    oh_b = torch.full((M, 32), 7.0, dtype=torch.bfloat16, device=dev)
    sentinel = torch.full_like(logits, -3.0)
    y = ops.conv(x, pw, bias, K, out=sentinel, post=dict(xt=lab_b, scalars=scal, K=K, E=E, philox_seed=991, philox_offset=off, draw=draw,
                                                        labels_out=lab_b, onehot_out=oh_b), **kw)
    assert y.fused_post                                       # really the fused epilogue
    assert torch.equal(sentinel, torch.full_like(logits, -3.0))          # the logits are NOT written in this mode
    assert torch.equal(lab_a, lab_b)
    assert torch.equal(oh_a, oh_b)
    assert torch.equal(oh_b[:, K:], torch.full((M, 32 - K), 7.0, dtype=torch.bfloat16, device=dev))      # channels >= K untouched
    assert len(torch.unique(lab_b)) >= 4                      # a non-trivial label field
    if draw:
        assert (lab_b != xt).float().mean() > 0.02


@pytest.mark.gpu
def test_fused_ccdm_reverse_step_is_refused_outside_its_envelope(dev):
    from jointimagegeneration_amd import ops
    K, Cin, sp = 14, 32, (8, 8, 16)                          # production dispatch: this grid runs on the box / gather kernels
    M = sp[0] * sp[1] * sp[2]
    x = ops.CL(torch.randn((1,) + sp + (Cin,), device=dev).bfloat16(), Cin)
    pw = ops.pack_conv_weight(torch.randn(K, Cin, 3, 3, 3, device=dev) * 0.05, Cin)
    xt = torch.zeros(M, dtype=torch.int32, device=dev)
    y = ops.conv(x, pw, None, K, k=(3, 3, 3), out_f32=True,
                 post=dict(xt=xt, scalars=torch.tensor([0.9, 0.5], device=dev), K=K, labels_out=xt, draw=False))
    assert not y.fused_post and torch.isfinite(y.t).all()    # plain conv: the caller launches the sampler kernel itself
