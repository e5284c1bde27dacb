"""Torch-facing wrappers over the C-ABI.  PyTorch only supplies device memory and the stream (plumbing);
every arithmetic op below is one or more hand-written HIP kernels in libguidegen_hip.so.

Activations travel as `CL` = channels-last bf16 tensors [N, D, H, W, Cpad] (D == 1 for 2-D), Cpad a multiple of 32
with zero pad lanes, plus the logical channel count.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import GG_BF16, GG_F32, AttentionDesc, ConvDesc, check


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def require_gpu(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{what}: tensor is on {t.device}; the GuideGen engine runs on MI355X only "
                           "(no CPU fallback; the CPU restatement lives in oracle/ and is test infrastructure)")


# fp32 VALIDATION mode (gg_f32.hip; include/guidegen_hip.h): inside `with ops.fp32_validation():` the layout movers create fp32
# channels-last tensors and fp32 [taps][Cin][Cout] weight packs, and every op below routes fp32 tensors to the fp32 kernels (fixed-order
# fp32 FMA accumulation, fp64 GroupNorm statistics, fp32 attention).  It exists so that the CCDM sampler's labels can be compared exactly
# with the fp32 reference; nothing on the production path is changed by it, and there is still no CPU path.
FP32 = False


class fp32_validation:
    def __enter__(self):
        global FP32
        self.prev, FP32 = FP32, True
        return self

    def __exit__(self, *exc):
        global FP32
        FP32 = self.prev
        return False


def is_f32(t) -> bool:
    return t.dtype == torch.float32


PATH_HINT = 0        # gg_conv_desc.path_hint: tests set 1 / 4 / 6 to run small shapes on the halo-tile kernel (production: 0)
def weights_token(module) -> Tuple[int, int, int]:
    """Cheap identity of a module's current weights: (#tensors, sum of in-place version counters, sum of storage addresses).
    load_state_dict, LitEma.copy_to / restore and optimizer steps write in place (version bump); .to(device) moves storage.
    NOT detected: writes through `p.data` (`p.data.copy_(...)`, as the reference's own ema.py:57-65 does) -- they bypass the
    version counter.  Code that writes that way must call `invalidate_caches(module)` afterwards."""
    n = v = a = 0
    for p in module.parameters():
        n += 1
        v += p._version
        a += p.data_ptr()
    return n, v, a


def invalidate_caches(module) -> None:
    """Bump the version counter of every parameter (an in-place no-op write), so that every cache keyed by `weights_token` or by
    (data_ptr, _version) -- repacked weights, time-bias tables, captured hipGraphs -- is rebuilt on the next forward.  For callers
    that changed weights through `p.data` (invisible to autograd's version counter)."""
    with torch.no_grad():
        for p in module.parameters():
            p.add_(0)


def capture_graph(fn) -> "torch.cuda.CUDAGraph":
    """Capture `fn()` (launches on the current stream) into a hipGraph.  The cyclic garbage collector must not run inside the
    capture: if it frees an older CUDAGraph object there, hipGraphDestroy is refused ("operation not permitted when stream is
    capturing") inside a destructor and the process aborts.  So: collect first, keep the collector off while capturing."""
    import gc
    g = torch.cuda.CUDAGraph()
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(g):
            fn()
    finally:
        if was_enabled:
            gc.enable()
    return g


def pad32(c: int) -> int:
    return (c + 31) // 32 * 32


@dataclass
class CL:
    """Channels-last activation: t is [N, D, H, W, Cpad] (bf16, or fp32 for head outputs); C = logical channels."""
    t: torch.Tensor
    C: int
    acc: Optional[torch.Tensor] = None      # int64 [N, stripes, Cpad, 2]: striped fixed-point per-channel (sum, sumsq) left by the producing conv
    fused_ddim: bool = False                # the producing (head) conv already applied the DDIM update (gg_conv_desc.ddim_x)
    fused_post: bool = False                # the producing (head) conv already ran the CCDM reverse step (gg_conv_desc.post_xt)

    @property
    def N(self): return self.t.shape[0]
    @property
    def spatial(self) -> Tuple[int, int, int]: return tuple(self.t.shape[1:4])
    @property
    def S(self) -> int: return self.t.shape[1] * self.t.shape[2] * self.t.shape[3]
    @property
    def Cpad(self) -> int: return self.t.shape[4]


# ----------------------------------------------------------------------------------------------- layout movers
def to_cl(x: torch.Tensor, c_pad: Optional[int] = None, out: Optional[torch.Tensor] = None, c_offset: int = 0,
          zero_fill: bool = True) -> CL:
    """NC[D]HW fp32 -> CL bf16 (optionally into channels [c_offset, c_offset+C) of an existing buffer)."""
    require_gpu(x, "to_cl")
    lib = _lib.load()
    x = x.contiguous().float()
    N, Cc = x.shape[:2]
    sp = tuple(x.shape[2:])
    sp3 = (1,) * (3 - len(sp)) + sp
    S = sp3[0] * sp3[1] * sp3[2]
    if (out is None and FP32) or (out is not None and is_f32(out)):      # validation mode: fp32 channels-last (layout copy = plumbing)
        if out is None:
            out = torch.zeros((N,) + sp3 + (c_pad or pad32(Cc + c_offset),), dtype=torch.float32, device=x.device)
        elif zero_fill:
            out.zero_()
        out[..., c_offset:c_offset + Cc] = x.reshape(N, Cc, S).permute(0, 2, 1).reshape((N,) + sp3 + (Cc,))
        return CL(out, Cc + c_offset)
    if out is None:
        cp = c_pad or pad32(Cc + c_offset)
        out = torch.empty((N,) + sp3 + (cp,), dtype=torch.bfloat16, device=x.device)
    cp = out.shape[-1]
    check(lib.gg_nchw_f32_to_cl_bf16(x.data_ptr(), N, Cc, S, out.data_ptr(), cp, c_offset, 1 if zero_fill else 0, _stream()),
          "gg_nchw_f32_to_cl_bf16")
    return CL(out, Cc + c_offset)


def from_cl(cl: CL, ndim_spatial: int) -> torch.Tensor:
    """CL (bf16 or fp32) -> NC[D]HW fp32."""
    lib = _lib.load()
    t = cl.t
    N, D, H, W, cp = t.shape
    sp = (D, H, W)[3 - ndim_spatial:]
    out = torch.empty((N, cl.C) + sp, dtype=torch.float32, device=t.device)
    dt = GG_BF16 if t.dtype == torch.bfloat16 else GG_F32
    check(lib.gg_cl_to_nchw_f32(t.data_ptr(), dt, N, cl.C, D * H * W, cp, out.data_ptr(), _stream()), "gg_cl_to_nchw_f32")
    return out


# ----------------------------------------------------------------------------------------------- conv
def pack_conv_weight(w: torch.Tensor, cin_pad: int) -> torch.Tensor:
    """fp32 OI[D]HW / OI (linear) -> MFMA tile order bf16 (device)."""
    require_gpu(w, "pack_conv_weight")
    lib = _lib.load()
    w = w.detach().contiguous().float()
    Cout, Cin = w.shape[:2]
    if FP32:                                     # validation mode: fp32 [taps][Cin_pad][Cout_pad], zero padded (layout copy = plumbing)
        wt = w.reshape(Cout, Cin, -1).permute(2, 1, 0)
        out = torch.zeros((wt.shape[0], cin_pad, pad32(Cout)), dtype=torch.float32, device=w.device)
        out[:, :Cin, :Cout] = wt
        return out
    ntaps = 1
    for s in w.shape[2:]:
        ntaps *= s
    nbytes = lib.gg_conv_packed_weight_bytes(Cout, cin_pad, ntaps)
    out = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=w.device)
    check(lib.gg_conv_pack_weight(w.data_ptr(), Cout, Cin, cin_pad, ntaps, out.data_ptr(), _stream()), "gg_conv_pack_weight")
    return out


def pad_bias(b: Optional[torch.Tensor], cout: int, device) -> torch.Tensor:
    out = torch.zeros(pad32(cout), dtype=torch.float32, device=device)
    if b is not None:
        out[:cout] = b.detach().float()
    return out


def conv_out_extent(in_sp: Sequence[int], k: Sequence[int], stride: int, pad: int, upsample: bool):
    out = []
    for s, kk in zip(in_sp, k):
        if kk == 1:
            out.append((s - 1) // stride + 1)
        else:
            e = s * 2 if upsample else s
            if stride == 1:
                out.append(e + 2 * pad - 2)
            else:
                out.append((e + 2 - 3) // 2 + 1 if pad == 1 else (e + 1 - 3) // 2 + 1)
    return tuple(out)


def conv_prologue_from_acc(src1: CL, cout: int, act: bool, k=(1, 3, 3), stride: int = 1, pad: int = 1, upsample: bool = False,
                           src2: Optional[CL] = None, **_ignored) -> bool:
    """True if this conv can compute act(GroupNorm(cat[src1, src2])) itself from the accumulators its producers left (CL.acc, 1-stripe
    layout): no statistics, scale / shift or apply launch at all (gg_conv_desc.pro_acc1)."""
    if not PROLOGUE_FROM_ACC or is_f32(src1.t) or not has_stats(src1, src2):
        return False
    if src2 is not None and src1.C != src1.Cpad:        # the in-kernel fold indexes channel c - C1 of the second source: C1 must be all logical
        return False
    lib = _lib.load()
    N, D, H, W, C1 = src1.t.shape
    Do, Ho, Wo = conv_out_extent((D, H, W), k, stride, pad, upsample)
    d = ConvDesc()
    d.N, d.D, d.H, d.W = N, D, H, W
    d.C1, d.C2 = C1, (src2.t.shape[-1] if src2 is not None else 0)
    d.Cout, d.Cout_pad = cout, pad32(cout)
    d.kd, d.kh, d.kw = k
    d.stride, d.pad, d.upsample = stride, pad, 1 if upsample else 0
    d.Do, d.Ho, d.Wo = Do, Ho, Wo
    d.prologue_act = 1 if act else 2
    d.pro_c_logical = src1.C + (src2.C if src2 is not None else 0)
    d.path_hint = PATH_HINT
    return bool(lib.gg_conv_prologue_from_acc(C.byref(d)))


SKIP_KCONCAT = True                 # ResBlock 1x1 skip projections K-concatenated into conv2 where gg_conv_fuses_skip says so (A/B switch)


def conv_fuses_skip(src1: CL, cout: int, skip1: CL, skip2: Optional[CL] = None, k=(1, 3, 3), stride: int = 1, pad: int = 1, upsample: bool = False,
                    **_ignored) -> bool:
    """True if this conv (input src1, no second source) can take a K-concatenated 1x1 skip projection of cat[skip1, skip2]
    (gg_conv_desc.skip_src1): box kernel, 3x3, stride 1."""
    if not SKIP_KCONCAT or is_f32(src1.t) or (skip2 is not None and skip1.C != skip1.Cpad):
        return False
    lib = _lib.load()
    N, D, H, W, C1 = src1.t.shape
    Do, Ho, Wo = conv_out_extent((D, H, W), k, stride, pad, upsample)
    d = ConvDesc()
    d.N, d.D, d.H, d.W = N, D, H, W
    d.C1, d.C2 = C1, 0
    d.Cout, d.Cout_pad = cout, pad32(cout)
    d.kd, d.kh, d.kw = k
    d.stride, d.pad, d.upsample = stride, pad, 1 if upsample else 0
    d.Do, d.Ho, d.Wo = Do, Ho, Wo
    d.skip_C1, d.skip_C2 = skip1.Cpad, (skip2.Cpad if skip2 is not None else 0)
    d.path_hint = PATH_HINT
    return bool(lib.gg_conv_fuses_skip(C.byref(d)))


def _shape_desc(src1: CL, cout: int, k, stride: int, pad: int, upsample: bool, src2: Optional[CL]) -> "ConvDesc":
    N, D, H, W, C1 = src1.t.shape
    Do, Ho, Wo = conv_out_extent((D, H, W), k, stride, pad, upsample)
    d = ConvDesc()
    d.N, d.D, d.H, d.W = N, D, H, W
    d.C1, d.C2 = C1, (src2.t.shape[-1] if src2 is not None else 0)
    d.Cout, d.Cout_pad = cout, pad32(cout)
    d.kd, d.kh, d.kw = k
    d.stride, d.pad, d.upsample = stride, pad, 1 if upsample else 0
    d.Do, d.Ho, d.Wo = Do, Ho, Wo
    d.path_hint = PATH_HINT
    return d


def conv_runs_halo_tile(src1: CL, cout: int, k=(1, 3, 3), stride: int = 1, pad: int = 1, upsample: bool = False,
                        src2: Optional[CL] = None, **_ignored) -> bool:
    """True if gg_conv_forward runs this shape on the halo-tile kernel."""
    if is_f32(src1.t):
        return False
    return bool(_lib.load().gg_conv_runs_halo_tile(C.byref(_shape_desc(src1, cout, k, stride, pad, upsample, src2))))


def conv_fuses_prologue(src1: CL, cout: int, k=(1, 3, 3), stride: int = 1, pad: int = 1, upsample: bool = False,
                        src2: Optional[CL] = None, **_ignored) -> bool:
    """True if the GroupNorm (* SiLU) in front of this conv should be handed to the conv as its prologue (applied once per element while
    staging); False: a separate apply pass + the prologue-free conv is faster (gg_conv_fuses_prologue, measured rule in gg_conv_halo.hip)."""
    if is_f32(src1.t):
        return False
    lib = _lib.load()
    N, D, H, W, C1 = src1.t.shape
    Do, Ho, Wo = conv_out_extent((D, H, W), k, stride, pad, upsample)
    d = ConvDesc()
    d.N, d.D, d.H, d.W = N, D, H, W
    d.C1, d.C2 = C1, (src2.t.shape[-1] if src2 is not None else 0)
    d.Cout, d.Cout_pad = cout, pad32(cout)
    d.kd, d.kh, d.kw = k
    d.stride, d.pad, d.upsample = stride, pad, 1 if upsample else 0
    d.Do, d.Ho, d.Wo = Do, Ho, Wo
    d.path_hint = PATH_HINT
    return bool(lib.gg_conv_fuses_prologue(C.byref(d)))


# ---- GroupNorm statistics emitted by conv epilogues (gg_conv_desc.gn_acc).  One int64 arena per device, bump-allocated per
# network forward and zeroed by ONE memset at the start of the next forward (static addresses: hipGraph friendly).
GN_ACC = True
TINY_IMAGE_POSITIONS = 0            # > 0: outputs with at most this many positions per sample also leave their sums, so that the next SiLU norm is folded into
                                    # its conv (16 = the 4x4 level: measured SLOWER, 1558 vs 1548 us per latent-UNet forward: producer epilogues + transform > launch)
# CCDM reverse step as the head conv's epilogue (gg_conv_desc.post_xt) where gg_conv_fuses_posterior says so (the 128^3 head).  Bit-identical
# to the two-launch form (tests).  What it buys is the 0.27 GB logit round trip only: the reverse step itself is ~3 k dependent vector
# instructions per voxel (IEEE divisions, Philox, exp) and VALU-bound wherever it runs -- rocprofv3, captured steps @128^3: head conv with the
# epilogue 523 us vs head conv 327 + sampler kernel 210 us (tools/experiments/probe_ccdm_step.py).  A/B switch.
FUSE_POSTERIOR = True
PROLOGUE_FROM_ACC = True            # box convs fold their producers' accumulators themselves where gg_conv_prologue_from_acc says so (A/B switch)
GN_ACC_MAX_ELEMS = 1 << 21          # per sample: only tensors whose norm is launch-bound (latent UNet at batch 1)
GN_ACC_MIN_ELEMS = 1 << 17          # below this the one-launch GroupNorm kernels are as fast (probe_gn_acc_min.py: 2^18 1593, 2^17 1586, 2^16 1590, 2^15 1604 us per forward)
_ARENA_ENTRIES = 1 << 19            # 4 MiB of int64 (the latent UNet at batch 1 uses ~0.4 M entries)
_ARENAS = {}


def stats_begin(device) -> None:
    """Start of a network forward: zero the arena and rewind.  Eager: zeroed is the prefix any forward so far has used (the high-water
    mark, rounded up to 64 Ki entries; entries beyond it have never been written: they are still the zeros of the allocation), so the
    fill (a dependent launch at the head of every forward, and 4 MiB of dirty L2 lines when the whole arena was zeroed) covers only what
    is in use (~1 MiB for the latent UNet).  While a hipGraph is being CAPTURED the whole arena is zeroed instead: the captured memset is
    fixed at capture time, and a forward captured cold (no eager run before it, so the mark still is 0) or one that allocates past the
    mark would otherwise replay onto its own previous sums."""
    if not GN_ACC:
        return
    a = _ARENAS.get(str(device))
    if a is None:
        a = _ARENAS[str(device)] = dict(buf=torch.zeros(_ARENA_ENTRIES, dtype=torch.int64, device=device), off=0, hi=0, active=False)
    else:
        a["hi"] = max(a["hi"], a["off"])
    if torch.cuda.is_current_stream_capturing():
        a["buf"].zero_()
        a["hi"] = _ARENA_ENTRIES                    # every later eager forward zeroes everything a replay may have dirtied
    elif a["hi"]:
        a["buf"][:min(_ARENA_ENTRIES, (a["hi"] + 65535) // 65536 * 65536)].zero_()
    a["off"] = 0
    a["active"] = True


def stats_end(device) -> None:
    a = _ARENAS.get(str(device))
    if a is not None:
        a["hi"] = max(a["hi"], a["off"])
        a["active"] = False


def _stats_alloc(device, N: int, cp: int, stripes: int) -> Optional[torch.Tensor]:
    a = _ARENAS.get(str(device))
    n = N * stripes * cp * 2            # stripes: 1 (box / 160-step kernels, GG_ACC_STRIPES) or 32 (halo-tile kernel)
    if a is None or not a["active"] or a["off"] + n > _ARENA_ENTRIES:
        return None
    v = a["buf"][a["off"]:a["off"] + n].view(N, stripes, cp, 2)
    a["off"] += n
    return v


def conv(src1: CL, weight: torch.Tensor, bias: Optional[torch.Tensor], cout: int, k=(1, 3, 3), stride: int = 1, pad: int = 1,
         upsample: bool = False, src2: Optional[CL] = None, residual: Optional[CL] = None, out_f32: bool = False,
         bias_per_sample: bool = False, prologue: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, prologue_silu: bool = True,
         out: Optional[torch.Tensor] = None, ddim: Optional[tuple] = None, geglu: bool = False, prologue_acc: Optional[tuple] = None,
         want_stats: bool = False, skip: Optional[tuple] = None, post: Optional[dict] = None) -> CL:
    """skip = (x1: CL, x2: CL or None, packed 1x1 weight): K-concatenated skip projection (conv_fuses_skip; `bias` must include its bias);
    prologue_acc = (gamma, beta, eps): GroupNorm prologue computed inside the conv from src1.acc / src2.acc (conv_prologue_from_acc);
    want_stats: leave the output's GroupNorm sums behind whatever its size (the consumer will fold them itself);
    ddim = (x fp32 [M,4], scalars fp32[4] on device, pred_x0 fp32 [M,4] or None, unet_in bf16 [M, stride] or None): the DDIM update
    runs as this (head) conv's epilogue when the kernel supports it (CL.fused_ddim tells); otherwise the caller launches gg_ddim_step;
    post = dict(xt, scalars, K, E, philox_seed, philox_offset, draw, labels_out, onehot_out): the CCDM reverse step (the arguments of
    ccdm_posterior_sample) as this (head) conv's epilogue where gg_conv_fuses_posterior says so (CL.fused_post tells; `out` is then not
    written); otherwise the caller launches gg_ccdm_posterior_sample on the logits."""
    lib = _lib.load()
    t1 = src1.t
    N, D, H, W, C1 = t1.shape
    Do, Ho, Wo = conv_out_extent((D, H, W), k, stride, pad, upsample)
    cp = pad32(cout)
    if is_f32(t1):                                  # fp32 validation path (gg_conv_forward_f32)
        if geglu or prologue is not None or not is_f32(weight) or (residual is not None and not is_f32(residual.t)):
            raise RuntimeError("fp32 validation conv: fp32 weights / residual, no fused prologue or GEGLU (enter ops.fp32_validation() before the first forward)")
        if out is None:
            out = torch.empty((N, Do, Ho, Wo, cp), dtype=torch.float32, device=t1.device)
        d = ConvDesc()
        d.N, d.D, d.H, d.W = N, D, H, W
        d.C1, d.C2 = C1, (src2.t.shape[-1] if src2 is not None else 0)
        d.Cout, d.Cout_pad = cout, cp
        d.kd, d.kh, d.kw = k
        d.stride, d.pad, d.upsample = stride, pad, 1 if upsample else 0
        d.Do, d.Ho, d.Wo = Do, Ho, Wo
        d.out_dtype = GG_F32
        d.src1, d.src2 = t1.data_ptr(), (_ptr(src2.t) if src2 is not None else None)
        d.weight, d.bias, d.bias_stride = weight.data_ptr(), _ptr(bias), (cp if bias_per_sample else 0)
        d.residual = _ptr(residual.t) if residual is not None else None
        d.out = out.data_ptr()
        check(lib.gg_conv_forward_f32(C.byref(d), _stream()), "gg_conv_forward_f32")
        return CL(out, cout)
    if geglu:      # fused GEGLU epilogue (gg_conv_desc.epilogue_geglu): cout = 2 * inner value | gate rows in, inner channels out
        if cout % 32 or out is not None or out_f32 or residual is not None:
            raise ValueError("conv(geglu=True): cout = 2 * inner with inner % 16 == 0, bf16 output, no residual")
        out = torch.empty((N, Do, Ho, Wo, cp // 2), dtype=torch.bfloat16, device=t1.device)
    if out is None:
        out = torch.empty((N, Do, Ho, Wo, cp), dtype=torch.float32 if out_f32 else torch.bfloat16, device=t1.device)
    d = ConvDesc()
    d.N, d.D, d.H, d.W = N, D, H, W
    d.C1, d.C2 = C1, (src2.t.shape[-1] if src2 is not None else 0)
    d.Cout, d.Cout_pad = cout, cp
    d.kd, d.kh, d.kw = k
    d.stride, d.pad, d.upsample = stride, pad, 1 if upsample else 0
    d.Do, d.Ho, d.Wo = Do, Ho, Wo
    d.out_dtype = GG_F32 if out.dtype == torch.float32 else GG_BF16
    d.prologue_act = (1 if prologue_silu else 2) if (prologue is not None or prologue_acc is not None) else 0
    if prologue_acc is not None:
        d.pro_gamma, d.pro_beta, d.pro_eps = prologue_acc[0].data_ptr(), prologue_acc[1].data_ptr(), float(prologue_acc[2])
        d.pro_acc1 = src1.acc.data_ptr()
        d.pro_acc2 = src2.acc.data_ptr() if src2 is not None else None
        d.pro_c_logical = src1.C + (src2.C if src2 is not None else 0)
    if skip is not None:
        d.skip_src1, d.skip_C1 = skip[0].t.data_ptr(), skip[0].Cpad
        d.skip_src2, d.skip_C2 = (skip[1].t.data_ptr(), skip[1].Cpad) if skip[1] is not None else (None, 0)
        d.skip_weight = skip[2].data_ptr()
    d.path_hint = PATH_HINT
    d.src1 = t1.data_ptr()
    d.src2 = _ptr(src2.t) if src2 is not None else None
    d.weight = weight.data_ptr()
    d.bias = _ptr(bias)
    d.bias_stride = cp if bias_per_sample else 0
    d.residual = _ptr(residual.t) if residual is not None else None
    d.out = out.data_ptr()
    d.gn_scale = _ptr(prologue[0]) if prologue is not None else None
    d.gn_shift = _ptr(prologue[1]) if prologue is not None else None
    d.epilogue_geglu = 1 if geglu else 0
    wsb = lib.gg_conv_workspace_bytes(C.byref(d))
    if wsb > 0:
        ws = torch.empty(wsb // 4, dtype=torch.float32, device=t1.device)
        d.workspace, d.workspace_bytes = ws.data_ptr(), wsb
    acc = None
    if GN_ACC and d.out_dtype == GG_BF16:
        # box / 160-step kernels (1 stripe): only where the norm is launch-bound; halo-tile kernel (32 stripes): always -- there the
        # sums replace a statistics PASS over a 17..805 MB tensor
        stripes = lib.gg_conv_emits_stats(C.byref(d))
        want = want_stats or (PROLOGUE_FROM_ACC and Do * Ho * Wo <= TINY_IMAGE_POSITIONS)
        if stripes == 32 or (stripes and (want or GN_ACC_MIN_ELEMS <= Do * Ho * Wo * cp) and Do * Ho * Wo * cp <= GN_ACC_MAX_ELEMS):
            acc = _stats_alloc(t1.device, N, cp, stripes)
            if acc is not None:
                d.gn_acc = acc.data_ptr()
    fused = False
    if ddim is not None and lib.gg_conv_fuses_ddim(C.byref(d)):
        x, scal, px0, uin = ddim
        d.ddim_x, d.ddim_scalars = x.data_ptr(), scal.data_ptr()
        d.ddim_pred_x0 = _ptr(px0)
        d.ddim_unet_in = _ptr(uin)
        d.ddim_unet_in_stride = uin.shape[-1] if uin is not None else 0
        fused = True
    fused_post = False
    if post is not None and FUSE_POSTERIOR and acc is None and post["K"] == cout and lib.gg_conv_fuses_posterior(C.byref(d)):
        oh = post.get("onehot_out")
        d.post_xt, d.post_labels_out, d.post_scalars = post["xt"].data_ptr(), post["labels_out"].data_ptr(), post["scalars"].data_ptr()
        d.post_E = _ptr(post.get("E"))
        d.post_philox_seed = int(post.get("philox_seed", 0))
        d.post_philox_offset_dev = _ptr(post.get("philox_offset"))
        d.post_draw = 1 if post.get("draw", True) else 0
        d.post_onehot_out, d.post_onehot_stride = _ptr(oh), (oh.shape[-1] if oh is not None else 0)
        fused_post = True
    check(lib.gg_conv_forward(C.byref(d), _stream()), "gg_conv_forward")
    return CL(out, cout // 2 if geglu else cout, acc=acc, fused_ddim=fused, fused_post=fused_post)


# ----------------------------------------------------------------------------------------------- norms / elementwise
def groupnorm_stats(src1: CL, gamma: torch.Tensor, beta: torch.Tensor, eps: float, src2: Optional[CL] = None):
    """Returns per-(n, c) fp32 (scale, shift) with y = x*scale + shift == GroupNorm(32, C)(x)."""
    lib = _lib.load()
    N, S = src1.N, src1.S
    C1 = src1.Cpad
    C2 = src2.Cpad if src2 is not None else 0
    c_log = src1.C + (src2.C if src2 is not None else 0)
    if src2 is not None and src1.C != C1:
        raise RuntimeError("two-source GroupNorm needs an unpadded first source")
    Ct = C1 + C2
    ws_bytes = lib.gg_groupnorm_workspace_bytes(N, S, Ct)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=src1.t.device)
    scale = torch.empty((N, Ct), dtype=torch.float32, device=src1.t.device)
    shift = torch.empty_like(scale)
    check(lib.gg_groupnorm_stats(src1.t.data_ptr(), C1, _ptr(src2.t) if src2 is not None else None, C2, N, S, c_log,
                                 gamma.data_ptr(), beta.data_ptr(), eps, scale.data_ptr(), shift.data_ptr(), ws.data_ptr(),
                                 ws_bytes, _stream()), "gg_groupnorm_stats")
    return scale, shift


def groupnorm_apply(src1: CL, scale: torch.Tensor, shift: torch.Tensor, act: bool, src2: Optional[CL] = None) -> CL:
    lib = _lib.load()
    N, S = src1.N, src1.S
    C1 = src1.Cpad
    C2 = src2.Cpad if src2 is not None else 0
    out = torch.empty(tuple(src1.t.shape[:4]) + (C1 + C2,), dtype=torch.bfloat16, device=src1.t.device)
    check(lib.gg_groupnorm_apply(src1.t.data_ptr(), C1, _ptr(src2.t) if src2 is not None else None, C2, N, S, scale.data_ptr(),
                                 shift.data_ptr(), 1 if act else 0, out.data_ptr(), _stream()), "gg_groupnorm_apply")
    return CL(out, src1.C + (src2.C if src2 is not None else 0))


def groupnorm_fused_ok(src1: CL, src2: Optional[CL] = None) -> bool:
    if is_f32(src1.t):
        return False
    C2 = src2.Cpad if src2 is not None else 0
    c_log = src1.C + (src2.C if src2 is not None else 0)
    if src2 is not None and src1.C != src1.Cpad:
        return False
    return bool(_lib.load().gg_groupnorm_fused_supported(src1.S, src1.Cpad, C2, c_log))


def groupnorm_fused(src1: CL, gamma: torch.Tensor, beta: torch.Tensor, eps: float, act: bool, src2: Optional[CL] = None) -> CL:
    """act(GroupNorm(32)(cat[src1, src2])) in one launch (small tensors: see groupnorm_fused_ok)."""
    lib = _lib.load()
    N, S = src1.N, src1.S
    C1 = src1.Cpad
    C2 = src2.Cpad if src2 is not None else 0
    c_log = src1.C + (src2.C if src2 is not None else 0)
    out = torch.empty(tuple(src1.t.shape[:4]) + (C1 + C2,), dtype=torch.bfloat16, device=src1.t.device)
    check(lib.gg_groupnorm_fused(src1.t.data_ptr(), C1, _ptr(src2.t) if src2 is not None else None, C2, N, S, c_log, gamma.data_ptr(),
                                 beta.data_ptr(), eps, 1 if act else 0, out.data_ptr(), _stream()), "gg_groupnorm_fused")
    return CL(out, c_log)


def groupnorm_f32(src1: CL, gamma: torch.Tensor, beta: torch.Tensor, eps: float, act: bool, src2: Optional[CL] = None) -> CL:
    """fp32 validation path: act(GroupNorm(32)(cat[src1, src2])) on fp32 CL tensors (fp64 statistics, ATen's fp32 affine order)."""
    lib = _lib.load()
    N, S = src1.N, src1.S
    C1 = src1.Cpad
    C2 = src2.Cpad if src2 is not None else 0
    c_log = src1.C + (src2.C if src2 is not None else 0)
    if src2 is not None and src1.C != C1:
        raise RuntimeError("two-source GroupNorm needs an unpadded first source")
    out = torch.empty(tuple(src1.t.shape[:4]) + (C1 + C2,), dtype=torch.float32, device=src1.t.device)
    ws = torch.empty(64 * N, dtype=torch.float32, device=src1.t.device)
    check(lib.gg_groupnorm_f32(src1.t.data_ptr(), C1, _ptr(src2.t) if src2 is not None else None, C2, N, S, c_log, gamma.data_ptr(), beta.data_ptr(),
                               eps, 1 if act else 0, out.data_ptr(), ws.data_ptr(), _stream()), "gg_groupnorm_f32")
    return CL(out, c_log)


def groupnorm_apply_acc(src1: CL, gamma: torch.Tensor, beta: torch.Tensor, eps: float, act: bool, src2: Optional[CL] = None) -> CL:
    """act(GroupNorm(32)(cat[src1, src2])) with the statistics taken from the accumulators the producing convs left in CL.acc."""
    lib = _lib.load()
    N, S = src1.N, src1.S
    C1 = src1.Cpad
    C2 = src2.Cpad if src2 is not None else 0
    c_log = src1.C + (src2.C if src2 is not None else 0)
    if src2 is not None and src1.C != C1:
        raise RuntimeError("two-source GroupNorm needs an unpadded first source")
    out = torch.empty(tuple(src1.t.shape[:4]) + (C1 + C2,), dtype=torch.bfloat16, device=src1.t.device)
    check(lib.gg_groupnorm_apply_acc(src1.t.data_ptr(), C1, src1.acc.data_ptr(), _ptr(src2.t) if src2 is not None else None, C2,
                                     src2.acc.data_ptr() if src2 is not None else None, N, S, c_log, gamma.data_ptr(),
                                     beta.data_ptr(), eps, 1 if act else 0, out.data_ptr(), _stream()), "gg_groupnorm_apply_acc")
    return CL(out, c_log)


def has_stats(src1: CL, src2: Optional[CL] = None) -> bool:
    """the producing convs (box / 160-step kernels) left their accumulators: gg_groupnorm_apply_acc can normalise without a statistics launch"""
    ok = lambda c: c is None or (c.acc is not None and c.acc.shape[1] != 32)       # box / 160-step producers (GG_ACC_STRIPES), not the halo kernel's 32
    return GN_ACC and src1.acc is not None and ok(src1) and ok(src2) and src1.Cpad + (src2.Cpad if src2 is not None else 0) <= 2048


def has_any_stats(src1: CL, src2: Optional[CL] = None) -> bool:
    """the producing convs left accumulators (any stripe count): gg_groupnorm_scale_shift_acc replaces the statistics pass"""
    return GN_ACC and src1.acc is not None and (src2 is None or src2.acc is not None)


def groupnorm_scale_shift_acc(src1: CL, gamma: torch.Tensor, beta: torch.Tensor, eps: float, src2: Optional[CL] = None):
    """Per-(n, c) fp32 (scale, shift) of GroupNorm(32)(cat[src1, src2]) from the accumulators in CL.acc (no pass over the tensors)."""
    lib = _lib.load()
    N, S = src1.N, src1.S
    C1 = src1.Cpad
    C2 = src2.Cpad if src2 is not None else 0
    c_log = src1.C + (src2.C if src2 is not None else 0)
    if src2 is not None and src1.C != C1:
        raise RuntimeError("two-source GroupNorm needs an unpadded first source")
    scale = torch.empty((N, C1 + C2), dtype=torch.float32, device=src1.t.device)
    shift = torch.empty_like(scale)
    check(lib.gg_groupnorm_scale_shift_acc(src1.acc.data_ptr(), src1.acc.shape[1], C1, src2.acc.data_ptr() if src2 is not None else None,
                                           src2.acc.shape[1] if src2 is not None else 0, C2, N, S, c_log, gamma.data_ptr(), beta.data_ptr(),
                                           eps, scale.data_ptr(), shift.data_ptr(), _stream()), "gg_groupnorm_scale_shift_acc")
    return scale, shift


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    lib = _lib.load()
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    out = torch.empty_like(x)
    check(lib.gg_layernorm(x.data_ptr(), rows, Cc, gamma.data_ptr(), beta.data_ptr(), eps, out.data_ptr(), _stream()), "gg_layernorm")
    return out


def geglu(h: torch.Tensor, inner: int) -> torch.Tensor:
    lib = _lib.load()
    rows = h.numel() // (2 * inner)
    out = torch.empty(tuple(h.shape[:-1]) + (inner,), dtype=torch.bfloat16, device=h.device)
    check(lib.gg_geglu(h.data_ptr(), rows, inner, out.data_ptr(), _stream()), "gg_geglu")
    return out


def add(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    out = torch.empty_like(a)
    check(lib.gg_add(a.data_ptr(), b.data_ptr(), a.numel(), out.data_ptr(), _stream()), "gg_add")
    return out


# ----------------------------------------------------------------------------------------------- attention
def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor, N: int, heads: int, head_dim: int, Tq: int,
              Tkv: int, ld_hs_q, ld_hs_k, ld_hs_v, ld_hs_o, scale: float, q_off=0, k_off=0, v_off=0) -> None:
    """q/k/v/out are bf16 tensors; element (n,t,h,d) at base + off + (n*T+t)*ld + h*hs + d."""
    lib = _lib.load()
    d = AttentionDesc()
    d.N, d.heads, d.head_dim, d.Tq, d.Tkv = N, heads, head_dim, Tq, Tkv
    d.ldq, d.hsq = ld_hs_q
    d.ldk, d.hsk = ld_hs_k
    d.ldv, d.hsv = ld_hs_v
    d.ldo, d.hso = ld_hs_o
    d.scale = scale
    esz = q.element_size()
    d.q = q.data_ptr() + esz * q_off
    d.k = k.data_ptr() + esz * k_off
    d.v = v.data_ptr() + esz * v_off
    d.out = out.data_ptr()
    if is_f32(q):                                   # fp32 validation path
        check(lib.gg_attention_forward_f32(C.byref(d), _stream()), "gg_attention_forward_f32")
        return
    wsb = lib.gg_attention_workspace_bytes(C.byref(d))
    if wsb > 0:                                     # under-filled single-head grid (AE mid attention): keys split over workgroups
        ws = torch.empty(wsb // 4, dtype=torch.float32, device=q.device)
        d.workspace, d.workspace_bytes = ws.data_ptr(), wsb
    check(lib.gg_attention_forward(C.byref(d), _stream()), "gg_attention_forward")


# ----------------------------------------------------------------------------------------------- small fp32 ops
def linear_f32(x: torch.Tensor, W: torch.Tensor, b: Optional[torch.Tensor], act_in: bool = False,
               out: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = _lib.load()
    x = x.contiguous()
    M, I = x.shape
    O = W.shape[0]
    if out is None:
        out = torch.empty((M, O), dtype=torch.float32, device=x.device)
    check(lib.gg_linear_f32(x.data_ptr(), M, I, W.data_ptr(), _ptr(b), O, 1 if act_in else 0, out.data_ptr(), out.stride(0), _stream()),
          "gg_linear_f32")
    return out


def timestep_embedding(t: torch.Tensor, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    lib = _lib.load()
    t = t.float().contiguous()
    out = torch.empty((t.shape[0], dim), dtype=torch.float32, device=t.device)
    check(lib.gg_timestep_embedding(t.data_ptr(), t.shape[0], dim, max_period, out.data_ptr(), _stream()), "gg_timestep_embedding")
    return out


# ----------------------------------------------------------------------------------------------- samplers
def ccdm_posterior_sample(head: torch.Tensor, head_is_logits: bool, xt: torch.Tensor, scalars: torch.Tensor, K: int, *,
                          E: Optional[torch.Tensor] = None, philox_seed: int = 0, philox_offset: Optional[torch.Tensor] = None,
                          draw: bool = True, labels_out: Optional[torch.Tensor] = None, probs_out: Optional[torch.Tensor] = None,
                          onehot_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """head: fp32 [..., stride] (probs or logits, channels-last); xt int32 [M]; scalars fp32[2] on device."""
    lib = _lib.load()
    stride = head.shape[-1]
    M = head.numel() // stride
    if labels_out is None:
        labels_out = torch.empty(M, dtype=torch.int32, device=head.device)
    check(lib.gg_ccdm_posterior_sample(head.data_ptr(), stride, 1 if head_is_logits else 0, xt.data_ptr(), _ptr(E), philox_seed,
                                       _ptr(philox_offset), 1 if draw else 0, scalars.data_ptr(), K, M, labels_out.data_ptr(),
                                       _ptr(probs_out), _ptr(onehot_out), onehot_out.shape[-1] if onehot_out is not None else 0,
                                       _stream()), "gg_ccdm_posterior_sample")
    return labels_out


def labels_to_onehot(labels: torch.Tensor, K: int, out: torch.Tensor) -> None:
    lib = _lib.load()
    if is_f32(out):                                 # fp32 validation path: index scatter (plumbing)
        out[:, :K] = torch.nn.functional.one_hot(labels.long(), K).to(torch.float32)
        return
    check(lib.gg_labels_to_onehot(labels.data_ptr(), labels.numel(), K, out.data_ptr(), out.shape[-1], _stream()), "gg_labels_to_onehot")


def ddim_step(x: torch.Tensor, eps: torch.Tensor, scalars: torch.Tensor, noise: Optional[torch.Tensor] = None,
              pred_x0_out: Optional[torch.Tensor] = None, unet_in: Optional[torch.Tensor] = None) -> None:
    """x fp32 CL [M, C] (updated in place); eps fp32 CL [M, stride]; scalars fp32[4] on device."""
    lib = _lib.load()
    Cc = x.shape[-1]
    M = x.numel() // Cc
    check(lib.gg_ddim_step(x.data_ptr(), eps.data_ptr(), eps.shape[-1], _ptr(noise), scalars.data_ptr(), M, Cc, _ptr(pred_x0_out),
                           _ptr(unet_in), unet_in.shape[-1] if unet_in is not None else 0, _stream()), "gg_ddim_step")


def minmax_normalise(src: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = _lib.load()
    if out is None:
        out = torch.empty_like(src)
    ws = torch.empty(2, dtype=torch.float32, device=src.device)
    check(lib.gg_minmax_normalise(src.data_ptr(), src.numel(), out.data_ptr(), ws.data_ptr(), _stream()), "gg_minmax_normalise")
    return out


def mask_to_cond_slice(labels: torch.Tensor, slice_idx: int, D: int, H: int, W: int, prev: Optional[torch.Tensor],
                       cond: torch.Tensor, mask_out: Optional[torch.Tensor] = None) -> None:
    """labels int32 [N,Dm,Hm,Wm] -> cond bf16 CL [N,1,H,W,stride] (ch0 prev slice, ch1 mask/255)."""
    lib = _lib.load()
    N, Dm, Hm, Wm = labels.shape
    check(lib.gg_mask_to_cond_slice(labels.data_ptr(), N, Dm, Hm, Wm, slice_idx, D, H, W, _ptr(prev), cond.data_ptr(), cond.shape[-1],
                                    _ptr(mask_out), _stream()), "gg_mask_to_cond_slice")


def zoom0_index(n_in: int, n_out: int) -> torch.Tensor:
    """Host-side index map of scipy.ndimage.zoom(order=0) along one axis (the rule gg_mask_to_cond_slice applies on the device):
    output o reads input floor(o * (n_in-1)/(n_out-1) + 0.5), IEEE double, product and sum rounded separately."""
    zf = (n_in - 1) / (n_out - 1) if n_out > 1 else 1.0
    o = torch.arange(n_out, dtype=torch.float64)
    return torch.clamp(torch.floor(o * zf + 0.5).long(), 0, n_in - 1)


def lincomb4(es, coefs, denom: float, out: torch.Tensor) -> torch.Tensor:
    """out = (sum_i coefs[i] * es[i]) / denom, fp32, left-to-right (PLMS multistep combination)."""
    lib = _lib.load()
    es = list(es) + [None] * (4 - len(es))
    cs = list(coefs) + [0.0] * (4 - len(coefs))
    check(lib.gg_lincomb4(es[0].data_ptr(), _ptr(es[1]), _ptr(es[2]), _ptr(es[3]), cs[0], cs[1], cs[2], cs[3], denom, out.numel(),
                          out.data_ptr(), _stream()), "gg_lincomb4")
    return out


def ddpm_step(x: torch.Tensor, eps: torch.Tensor, scalars: torch.Tensor, noise: Optional[torch.Tensor] = None,
              unet_in: Optional[torch.Tensor] = None) -> None:
    """Ancestral DDPM update in place; x fp32 CL [M, C], eps fp32 CL [M, stride], scalars fp32[5] on device."""
    lib = _lib.load()
    Cc = x.shape[-1]
    M = x.numel() // Cc
    check(lib.gg_ddpm_step(x.data_ptr(), eps.data_ptr(), eps.shape[-1], _ptr(noise), scalars.data_ptr(), M, Cc, _ptr(unet_in),
                           unet_in.shape[-1] if unet_in is not None else 0, _stream()), "gg_ddpm_step")
