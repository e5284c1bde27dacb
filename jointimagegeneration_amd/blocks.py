"""Network blocks of the two GuideGen UNets and the KL autoencoder, executed by the HIP engine.

torch.nn.{ConvNd, GroupNorm, Linear, LayerNorm} objects appear here ONLY as parameter containers, so that
`state_dict()` has exactly the reference's names and shapes (SURVEY.md 8b "state-dict surface"); their ATen
`forward` is never called.  Each block's `run(...)` enqueues hand-written HIP kernels through `ops`.

Reference blocks mirrored (file:line relative to the reference tree):
  ResBlock / Upsample / Downsample / AttentionBlock / TimestepEmbedSequential
      ccdm/ddpm/models/unet_openai/unet.py:70-360, latentdiffusion/ldm/modules/diffusionmodules/openaimodel.py:74-375
  SpatialTransformer / BasicTransformerBlock / CrossAttention / GEGLU
      latentdiffusion/ldm/modules/attention.py:37-64,152-261
  ResnetBlock / AttnBlock2d / Upsample / Downsample (AE)
      latentdiffusion/ldm/modules/diffusionmodules/model.py:42-145,209-261
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch
import torch.nn as nn

from . import ops
from .ops import CL, pad32


def conv_nd(dims, *a, **k):
    return {1: nn.Conv1d, 2: nn.Conv2d, 3: nn.Conv3d}[dims](*a, **k)


def zero_module(m):
    """Reference initialisation of the last conv of each block (nn.py:68-74); only matters for fresh models."""
    for p in m.parameters():
        p.detach().zero_()
    return m


def _k3(w: torch.Tensor):
    """(kd, kh, kw) of a conv weight; 1-D and 2-D kernels are left-padded with 1s."""
    ks = tuple(w.shape[2:])
    return (1,) * (3 - len(ks)) + ks


class _Packed:
    """Cache of kernel-friendly repacks (bf16 MFMA tile order, padded fp32 biases) keyed by parameter version."""

    def __init__(self):
        self.store = {}

    def get(self, key, params, build):
        ver = tuple((p.data_ptr(), p._version) for p in params if p is not None)
        hit = self.store.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        val = build()
        self.store[key] = (ver, val)
        return val


PACKED = _Packed()


def packed_conv(mod: nn.Module, cin_pad: int, tag: str = ""):
    """(packed bf16 weight, padded fp32 bias) of a conv/linear container for inputs with `cin_pad` channels."""
    def build():
        w = mod.weight
        if w.ndim == 2:
            w = w[:, :, None]
        pw = ops.pack_conv_weight(w, cin_pad)
        pb = ops.pad_bias(getattr(mod, "bias", None), w.shape[0], w.device)
        return pw, pb
    return PACKED.get((id(mod), cin_pad, tag, ops.FP32), [mod.weight, getattr(mod, "bias", None)], build)


def packed_cat(mods: List[nn.Module], cin_pad: int, tag: str):
    """Several convs/linears sharing an input, fused along Cout (q|k|v projections)."""
    def build():
        ws = [m.weight if m.weight.ndim > 2 else m.weight[:, :, None] for m in mods]
        ws = [w.reshape(w.shape[0], w.shape[1], -1) for w in ws]
        w = torch.cat(ws, 0)
        bs = [m.bias if getattr(m, "bias", None) is not None else torch.zeros(m.weight.shape[0], device=w.device) for m in mods]
        pw = ops.pack_conv_weight(w, cin_pad)
        pb = ops.pad_bias(torch.cat(bs, 0), w.shape[0], w.device)
        return pw, pb
    params = [m.weight for m in mods] + [getattr(m, "bias", None) for m in mods]
    return PACKED.get((id(mods[0]), cin_pad, tag, ops.FP32), params, build)


def packed_geglu(proj: nn.Module, cin_pad: int):
    """GEGLU projection (attention.py:37-44) packed for the fused epilogue (gg_conv_desc.epilogue_geglu): rows in groups of 32 =
    [16 value rows j0..j0+15 | their 16 gate rows inner+j0..inner+j0+15], bias likewise."""
    def build():
        w, b = proj.weight, proj.bias
        inner = w.shape[0] // 2
        assert inner % 16 == 0
        j = torch.arange(inner, device=w.device).view(-1, 16)                       # [inner/16, 16] value rows
        order = torch.cat([j, j + inner], 1).reshape(-1)                            # value block, gate block, value block, ...
        pw = ops.pack_conv_weight(w[order][:, :, None], cin_pad)
        pb = ops.pad_bias(b[order] if b is not None else None, w.shape[0], w.device)
        return pw, pb
    return PACKED.get((id(proj), cin_pad, "geglu"), [proj.weight, getattr(proj, "bias", None)], build)


FUSE_GEGLU = True     # feed-forward projection with the GEGLU as its epilogue (False: projection, then gg_geglu)


def f32(p: torch.Tensor) -> torch.Tensor:
    return p.detach().float().contiguous()


def gn_silu(h: CL, norm: nn.GroupNorm, act: bool, src2: Optional[CL] = None) -> CL:
    if ops.is_f32(h.t):                      # fp32 validation path
        return ops.groupnorm_f32(h, f32(norm.weight), f32(norm.bias), norm.eps, act, src2)
    if ops.groupnorm_fused_ok(h, src2):     # small tensor: statistics + apply in one launch
        return ops.groupnorm_fused(h, f32(norm.weight), f32(norm.bias), norm.eps, act, src2)
    if ops.has_stats(h, src2):      # the producing convs already left the per-channel sums: no statistics launch
        return ops.groupnorm_apply_acc(h, f32(norm.weight), f32(norm.bias), norm.eps, act, src2)
    scale, shift = ops.groupnorm_stats(h, f32(norm.weight), f32(norm.bias), norm.eps, src2)
    return ops.groupnorm_apply(h, scale, shift, act, src2)


def norm_conv(h: CL, norm: nn.GroupNorm, act: bool, weight, bias, cout, src2: Optional[CL] = None, **conv_kw) -> CL:
    """conv(act(GroupNorm(cat[h, src2]))).
    Halo-tile convs (3x3(x3), stride 1, large extents): one stats pass, then normalise*affine(+SiLU) and the skip concat are
    fused into the conv's staging pass (applied once per staged element) -- the activation is never re-written to HBM.
    Gather-kernel convs: separate apply pass (measured: SiLU inside the latency-bound gather loop costs 26 vs 16.6 us/conv)."""
    if ops.is_f32(h.t):                      # fp32 validation path: separate fp32 GroupNorm launch, then the fp32 conv
        return ops.conv(ops.groupnorm_f32(h, f32(norm.weight), f32(norm.bias), norm.eps, act, src2), weight, bias, cout, **conv_kw)
    if ops.conv_prologue_from_acc(h, cout, act, src2=src2, **conv_kw):
        # box conv + producers' sums: the conv folds them and normalises its staged box itself -- NO GroupNorm launch of any kind
        return ops.conv(h, weight, bias, cout, src2=src2, prologue_acc=(f32(norm.weight), f32(norm.bias), norm.eps), prologue_silu=act, **conv_kw)
    fused = ops.conv_fuses_prologue(h, cout, src2=src2, **conv_kw)
    if not fused and ops.groupnorm_fused_ok(h, src2):     # small tensor: statistics + apply in ONE launch
        a = ops.groupnorm_fused(h, f32(norm.weight), f32(norm.bias), norm.eps, act, src2)
        return ops.conv(a, weight, bias, cout, **conv_kw)
    if not fused and ops.has_stats(h, src2):     # statistics came with the tensor (conv epilogue accumulators)
        a = ops.groupnorm_apply_acc(h, f32(norm.weight), f32(norm.bias), norm.eps, act, src2)
        return ops.conv(a, weight, bias, cout, **conv_kw)
    if ops.has_any_stats(h, src2):     # halo-tile producers: fold their sums instead of re-reading 17..805 MB per norm
        scale, shift = ops.groupnorm_scale_shift_acc(h, f32(norm.weight), f32(norm.bias), norm.eps, src2)
    else:
        scale, shift = ops.groupnorm_stats(h, f32(norm.weight), f32(norm.bias), norm.eps, src2)
    if fused:
        return ops.conv(h, weight, bias, cout, src2=src2, prologue=(scale, shift), prologue_silu=act, **conv_kw)
    a = ops.groupnorm_apply(h, scale, shift, act, src2)
    return ops.conv(a, weight, bias, cout, **conv_kw)


class TimestepBlock(nn.Module):
    pass


class GroupNorm32(nn.GroupNorm):
    """Parameter container; statistics are fp32 in the HIP kernel as in the reference (nn.py:17-19)."""


def normalization(ch):
    return GroupNorm32(32, ch)


# ------------------------------------------------------------------------------------------------ UNet blocks
class Upsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None):
        super().__init__()
        self.channels, self.out_channels, self.use_conv, self.dims = channels, out_channels or channels, use_conv, dims
        if not use_conv:
            raise NotImplementedError("Upsample without conv is not on the scoped path (conv_resample=True everywhere)")
        self.conv = conv_nd(dims, channels, self.out_channels, 3, padding=1)

    def run(self, h: CL) -> CL:
        pw, pb = packed_conv(self.conv, h.Cpad)
        return ops.conv(h, pw, pb, self.out_channels, k=_k3(self.conv.weight), stride=1, pad=1, upsample=True)


class Downsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None):
        super().__init__()
        self.channels, self.out_channels, self.use_conv, self.dims = channels, out_channels or channels, use_conv, dims
        if not use_conv:
            raise NotImplementedError("avg-pool Downsample is not on the scoped path (conv_resample=True everywhere)")
        self.op = conv_nd(dims, channels, self.out_channels, 3, stride=2, padding=1)

    def run(self, h: CL) -> CL:
        pw, pb = packed_conv(self.op, h.Cpad)
        # the output is a skip tensor: a decoder GroupNorm over cat[h, skip] takes its statistics from accumulators only if BOTH sources
        # carry them, so leave the sums behind also below ops.GN_ACC_MIN_ELEMS (the 16x16 x 320 tensor of the latent UNet: otherwise
        # that norm falls back to a statistics launch + an apply launch)
        small = (h.S // (4 if self.dims == 2 else 8)) * pad32(self.out_channels) >= (1 << 16)
        return ops.conv(h, pw, pb, self.out_channels, k=_k3(self.op.weight), stride=2, pad=1, want_stats=small)


class ResBlock(TimestepBlock):
    """GN+SiLU -> conv3 (+timestep bias) -> GN+SiLU -> conv3 (+skip).  The timestep projection
    `emb_layers` is folded into conv1's per-sample bias (SURVEY.md 2.3 "timestep-embed epilogue")."""

    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_conv=False, use_scale_shift_norm=False,
                 dims=2, use_checkpoint=False, up=False, down=False):
        super().__init__()
        if use_scale_shift_norm or up or down:
            raise NotImplementedError("use_scale_shift_norm / resblock_updown are not used by any shipped config")
        self.channels, self.emb_channels = channels, emb_channels
        self.out_channels = out_channels or channels
        self.in_layers = nn.Sequential(normalization(channels), nn.SiLU(), conv_nd(dims, channels, self.out_channels, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_channels, self.out_channels))
        self.out_layers = nn.Sequential(normalization(self.out_channels), nn.SiLU(), nn.Dropout(p=dropout),
                                        zero_module(conv_nd(dims, self.out_channels, self.out_channels, 3, padding=1)))
        if self.out_channels == channels:
            self.skip_connection = nn.Identity()
        elif use_conv:
            self.skip_connection = conv_nd(dims, channels, self.out_channels, 3, padding=1)
        else:
            self.skip_connection = conv_nd(dims, channels, self.out_channels, 1)

    def time_bias(self, emb: torch.Tensor, out: torch.Tensor) -> None:
        """out[M, Cout_pad] = conv1.bias + Linear(SiLU(emb))  (unet.py:251-260)."""
        lin, c1 = self.emb_layers[1], self.in_layers[2]
        b = PACKED.get((id(self), "tb"), [lin.bias, c1.bias], lambda: (f32(lin.bias) + f32(c1.bias)))
        ops.linear_f32(emb, f32(lin.weight), b, act_in=True, out=out)

    def run(self, h: CL, tbias: torch.Tensor, src2: Optional[CL] = None, want_stats: bool = False) -> CL:
        """want_stats: the consumer of this block's output folds GroupNorm sums itself (an AttentionBlock's norm in front of its box-kernel
        qkv conv, or a tiny-image ResBlock): conv2 leaves them even for tensors below ops.GN_ACC_MIN_ELEMS."""
        c1, c2 = self.in_layers[2], self.out_layers[3]
        k = _k3(c1.weight)
        cin_pad = h.Cpad + (src2.Cpad if src2 is not None else 0)
        # (running the 1x1 skip projection on a side stream beside the GN -> conv1 -> GN chain was measured slower: the fork / join
        # edges in the hipGraph cost more than the overlap saves, 3.03 vs 2.69 ms per latent-UNet forward)
        res, skip_kw = h, {}
        pw1, _ = packed_conv(c1, cin_pad)
        pw2, pb2 = packed_conv(c2, pad32(self.out_channels))
        sk = self.skip_connection
        if not isinstance(sk, nn.Identity):
            pws, pbs = packed_conv(sk, cin_pad)
            ks = _k3(sk.weight)
            h1_shape = CL(h.t[..., :1].expand(*h.t.shape[:4], pad32(self.out_channels)), self.out_channels)     # shape carrier for the query (no data read)
            if ks == (1, 1, 1) and ops.conv_fuses_skip(h1_shape, self.out_channels, h, src2, k=k):
                # 1x1 skip projection K-concatenated into conv2 (gg_conv_desc.skip_src1): no skip launch, no residual round trip
                pb_c2 = pb2
                pb2 = PACKED.get((id(self), "b2s"), [c2.bias, sk.bias], lambda: pb_c2 + pbs)          # conv2 bias + skip bias
                res, skip_kw = None, {"skip": (h, src2, pws)}
            else:
                res = ops.conv(h, pws, pbs, self.out_channels, k=ks, pad=ks[-1] // 2, src2=src2)
        h1 = norm_conv(h, self.in_layers[0], True, pw1, tbias, self.out_channels, src2=src2, k=k, bias_per_sample=True)
        assert h1.Cpad == pad32(self.out_channels)
        return norm_conv(h1, self.out_layers[0], True, pw2, pb2, self.out_channels, k=k, residual=res, want_stats=want_stats, **skip_kw)


class AttentionBlock(nn.Module):
    """GN -> qkv 1x1 -> QKVAttentionLegacy -> proj 1x1 + x, with flash attention tiles (no TxT buffer)."""

    def __init__(self, channels, num_heads=1, num_head_channels=-1, use_checkpoint=False, use_new_attention_order=False):
        super().__init__()
        if use_new_attention_order:
            raise NotImplementedError("use_new_attention_order is not used by any shipped config")
        self.channels = channels
        if num_head_channels == -1:
            self.num_heads = num_heads
        else:
            assert channels % num_head_channels == 0, \
                f"q,k,v channels {channels} is not divisible by num_head_channels {num_head_channels}"
            self.num_heads = channels // num_head_channels
        self.norm = normalization(channels)
        self.qkv = conv_nd(1, channels, channels * 3, 1)
        self.proj_out = zero_module(conv_nd(1, channels, channels, 1))

    def run(self, h: CL) -> CL:
        Cc, nh = self.channels, self.num_heads
        ch = Cc // nh
        N, T = h.N, h.S
        pw, pb = packed_conv(self.qkv, h.Cpad)
        qkv = norm_conv(h, self.norm, False, pw, pb, 3 * Cc, k=(1, 1, 1), pad=0)   # legacy order: head-major, q|k|v per head
        att = torch.empty(tuple(h.t.shape[:4]) + (Cc,), dtype=h.t.dtype, device=h.t.device)
        ld = qkv.Cpad
        ops.attention(qkv.t, qkv.t, qkv.t, att, N, nh, ch, T, T, (ld, 3 * ch), (ld, 3 * ch), (ld, 3 * ch), (Cc, ch),
                      1.0 / math.sqrt(ch), q_off=0, k_off=ch, v_off=2 * ch)
        pw2, pb2 = packed_conv(self.proj_out, Cc)
        return ops.conv(CL(att, Cc), pw2, pb2, Cc, k=(1, 1, 1), pad=0, residual=h)


# ------------------------------------------------------------------------------------------------ SpatialTransformer
class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, glu=True, dropout=0.0):
        super().__init__()
        if not glu:
            raise NotImplementedError("non-gated FeedForward is not used by BasicTransformerBlock")
        inner = int(dim * mult)
        self.inner = inner
        self.net = nn.Sequential(GEGLU(dim, inner), nn.Dropout(dropout), nn.Linear(inner, dim_out or dim))


class CrossAttention(nn.Module):
    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.0):
        super().__init__()
        inner = dim_head * heads
        self.heads, self.dim_head, self.inner = heads, dim_head, inner
        self.scale = dim_head ** -0.5
        context_dim = context_dim or query_dim
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(context_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(dropout))

    def run(self, xn: CL, context: Optional[CL], residual: CL) -> CL:
        """to_out(attn(to_q(xn), to_k(ctx), to_v(ctx))) + residual; xn/context are token rows as CL [N,1,1,T,C]."""
        N, T = xn.N, xn.S
        inner, hd = self.inner, self.dim_head
        if context is None:
            pw, pb = packed_cat([self.to_q, self.to_k, self.to_v], xn.Cpad, "qkv")
            qkv = ops.conv(xn, pw, pb, 3 * inner, k=(1, 1, 1), pad=0)
            q = k = v = qkv.t
            ldq = ldk = qkv.Cpad
            qo, ko, vo, Tkv = 0, inner, 2 * inner, T
        else:
            pwq, pbq = packed_conv(self.to_q, xn.Cpad)
            qc = ops.conv(xn, pwq, pbq, inner, k=(1, 1, 1), pad=0)
            pwk, pbk = packed_cat([self.to_k, self.to_v], context.Cpad, "kv")
            kvc = ops.conv(context, pwk, pbk, 2 * inner, k=(1, 1, 1), pad=0)
            q, k, v = qc.t, kvc.t, kvc.t
            ldq, ldk = qc.Cpad, kvc.Cpad
            qo, ko, vo, Tkv = 0, 0, inner, context.S
        att = torch.empty(tuple(xn.t.shape[:4]) + (inner,), dtype=torch.bfloat16, device=xn.t.device)
        ops.attention(q, k, v, att, N, self.heads, hd, T, Tkv, (ldq, hd), (ldk, hd), (ldk, hd), (inner, hd), self.scale,
                      q_off=qo, k_off=ko, v_off=vo)
        pwo, pbo = packed_conv(self.to_out[0], inner)
        return ops.conv(CL(att, inner), pwo, pbo, self.to_out[0].weight.shape[0], k=(1, 1, 1), pad=0, residual=residual)


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, n_heads, d_head, dropout=0.0, context_dim=None, gated_ff=True, checkpoint=True):
        super().__init__()
        self.attn1 = CrossAttention(query_dim=dim, heads=n_heads, dim_head=d_head, dropout=dropout)
        self.ff = FeedForward(dim, dropout=dropout, glu=gated_ff)
        self.attn2 = CrossAttention(query_dim=dim, context_dim=context_dim, heads=n_heads, dim_head=d_head, dropout=dropout)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(dim), nn.LayerNorm(dim), nn.LayerNorm(dim)

    def _ln(self, x: CL, ln: nn.LayerNorm) -> CL:
        return CL(ops.layernorm(x.t, f32(ln.weight), f32(ln.bias), ln.eps), x.C)

    def run(self, x: CL, context: Optional[CL]) -> CL:
        x = self.attn1.run(self._ln(x, self.norm1), None, x)
        x = self.attn2.run(self._ln(x, self.norm2), context, x)
        y = self._ln(x, self.norm3)
        proj, lin2 = self.ff.net[0].proj, self.ff.net[2]
        if FUSE_GEGLU and self.ff.inner % 16 == 0:
            pw, pb = packed_geglu(proj, y.Cpad)
            ggcl = ops.conv(y, pw, pb, proj.weight.shape[0], k=(1, 1, 1), pad=0, geglu=True)      # value * gelu(gate) from the fp32 accumulators
        else:
            pw, pb = packed_conv(proj, y.Cpad)
            hcl = ops.conv(y, pw, pb, proj.weight.shape[0], k=(1, 1, 1), pad=0)
            ggcl = CL(ops.geglu(hcl.t, self.ff.inner), self.ff.inner)
        pw2, pb2 = packed_conv(lin2, self.ff.inner)
        return ops.conv(ggcl, pw2, pb2, lin2.weight.shape[0], k=(1, 1, 1), pad=0, residual=x)


class SpatialTransformer(nn.Module):
    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0.0, context_dim=None):
        super().__init__()
        self.in_channels = in_channels
        inner = n_heads * d_head
        self.inner = inner
        self.norm = nn.GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Conv2d(in_channels, inner, kernel_size=1, stride=1, padding=0)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, n_heads, d_head, dropout=dropout, context_dim=context_dim) for _ in range(depth)])
        self.proj_out = zero_module(nn.Conv2d(inner, in_channels, kernel_size=1, stride=1, padding=0))

    def run(self, h: CL, context: Optional[CL]) -> CL:
        pw, pb = packed_conv(self.proj_in, h.Cpad)
        x = norm_conv(h, self.norm, False, pw, pb, self.inner, k=(1, 1, 1), pad=0)   # 'b c h w -> b (h w) c' is free in CL
        for blk in self.transformer_blocks:
            x = blk.run(x, context)
        pw2, pb2 = packed_conv(self.proj_out, x.Cpad)
        return ops.conv(x, pw2, pb2, self.in_channels, k=(1, 1, 1), pad=0, residual=h)


class TimestepEmbedSequential(nn.Sequential, TimestepBlock):
    """Dispatches (h, time-bias, context, skip) to the children that take them (unet.py:70-84)."""

    def run(self, h: CL, tbias_of, context: Optional[CL], skip: Optional[CL] = None) -> CL:
        layers = list(self)
        for i, layer in enumerate(layers):
            if isinstance(layer, ResBlock):
                # the next layer's norm can be folded into its (box-kernel) conv from this block's output sums
                nxt = layers[i + 1] if i + 1 < len(layers) else None
                h = layer.run(h, tbias_of(layer), skip, want_stats=isinstance(nxt, (AttentionBlock, SpatialTransformer)))
                skip = None
            elif isinstance(layer, SpatialTransformer):
                h = layer.run(h, context)
            elif isinstance(layer, (AttentionBlock, Upsample, Downsample)):
                h = layer.run(h)
            elif isinstance(layer, (nn.Conv1d, nn.Conv2d, nn.Conv3d)):
                pw, pb = packed_conv(layer, h.Cpad)
                h = ops.conv(h, pw, pb, layer.weight.shape[0], k=_k3(layer.weight), pad=1)
            else:
                raise TypeError(f"no HIP execution rule for {type(layer).__name__}")
        if skip is not None:
            raise RuntimeError("skip tensor was not consumed by a ResBlock")
        return h


# ------------------------------------------------------------------------------------------------ AE blocks
def Normalize(in_channels, num_groups=32):
    return nn.GroupNorm(num_groups=num_groups, num_channels=in_channels, eps=1e-6, affine=True)


class AEUpsample(nn.Module):
    def __init__(self, in_channels, with_conv, dims=2):
        super().__init__()
        assert with_conv and dims == 2
        self.with_conv = with_conv
        self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)

    def run(self, h: CL) -> CL:
        pw, pb = packed_conv(self.conv, h.Cpad)
        return ops.conv(h, pw, pb, self.conv.weight.shape[0], k=(1, 3, 3), upsample=True)


class AEDownsample(nn.Module):
    """pad (0,1,0,1) then stride-2 valid conv (model.py:75-79) = conv with leading pad 0, trailing pad 1."""

    def __init__(self, in_channels, with_conv, dims=2):
        super().__init__()
        assert with_conv and dims == 2
        self.with_conv = with_conv
        self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=2, padding=0)

    def run(self, h: CL) -> CL:
        pw, pb = packed_conv(self.conv, h.Cpad)
        return ops.conv(h, pw, pb, self.conv.weight.shape[0], k=(1, 3, 3), stride=2, pad=0)


class ResnetBlock(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=0, dims=2):
        super().__init__()
        assert dims == 2 and temb_channels == 0 and not conv_shortcut
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1 = Normalize(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, 1, 1)
        self.norm2 = Normalize(out_channels)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, 1, 1)
        if in_channels != out_channels:
            self.nin_shortcut = nn.Conv2d(in_channels, out_channels, 1, 1, 0)

    def run(self, h: CL) -> CL:
        pw1, pb1 = packed_conv(self.conv1, h.Cpad)
        h1 = norm_conv(h, self.norm1, True, pw1, pb1, self.out_channels, k=(1, 3, 3))
        res = h
        if self.in_channels != self.out_channels:
            pws, pbs = packed_conv(self.nin_shortcut, h.Cpad)
            res = ops.conv(h, pws, pbs, self.out_channels, k=(1, 1, 1), pad=0)
        pw2, pb2 = packed_conv(self.conv2, h1.Cpad)
        return norm_conv(h1, self.norm2, True, pw2, pb2, self.out_channels, k=(1, 3, 3), residual=res)


class AttnBlock2d(nn.Module):
    """Single-head attention over c channels, scale c^-1/2 (model.py:209-261); q|k|v fused into one 1x1 conv."""

    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = nn.Conv2d(in_channels, in_channels, 1, 1, 0)
        self.k = nn.Conv2d(in_channels, in_channels, 1, 1, 0)
        self.v = nn.Conv2d(in_channels, in_channels, 1, 1, 0)
        self.proj_out = nn.Conv2d(in_channels, in_channels, 1, 1, 0)

    def run(self, h: CL) -> CL:
        Cc = self.in_channels
        N, T = h.N, h.S
        pw, pb = packed_cat([self.q, self.k, self.v], h.Cpad, "qkv")
        qkv = norm_conv(h, self.norm, False, pw, pb, 3 * Cc, k=(1, 1, 1), pad=0)
        att = torch.empty(tuple(h.t.shape[:4]) + (Cc,), dtype=torch.bfloat16, device=h.t.device)
        ld = qkv.Cpad
        ops.attention(qkv.t, qkv.t, qkv.t, att, N, 1, Cc, T, T, (ld, Cc), (ld, Cc), (ld, Cc), (Cc, Cc), int(Cc) ** -0.5,
                      q_off=0, k_off=Cc, v_off=2 * Cc)
        pw2, pb2 = packed_conv(self.proj_out, Cc)
        return ops.conv(CL(att, Cc), pw2, pb2, Cc, k=(1, 1, 1), pad=0, residual=h)
