"""The two GuideGen UNets (CCDM 3-D categorical UNet, LDM 2-D UNetModel) on the HIP engine.

Constructor arguments, attribute names and state_dict keys follow the reference so configs and checkpoints drop in:
  CCDM  ccdm/ddpm/models/unet_openai/unet.py:402-823, ccdm/ddpm/models/unet_openai/__init__.py:5-66
  LDM   latentdiffusion/ldm/modules/diffusionmodules/openaimodel.py:416-745
Both share one block-layout routine (`_layout`); execution is channels-last bf16 through `blocks.*.run`.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import ops
from .blocks import (AttentionBlock, Downsample, ResBlock, SpatialTransformer, TimestepEmbedSequential, Upsample, conv_nd,
                     f32, gn_silu, norm_conv, normalization, packed_conv, zero_module, _k3)
from .ops import CL, pad32


def _layout(self, *, dims, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions, dropout,
            channel_mult, conv_resample, num_heads, num_head_channels, num_heads_upsample, make_attention):
    """Builds time_embed / input_blocks / middle_block / output_blocks with the reference's indices."""
    ted = model_channels * 4
    self.time_embed = nn.Sequential(nn.Linear(model_channels, ted), nn.SiLU(), nn.Linear(ted, ted))
    ch = int(channel_mult[0] * model_channels)
    first = ch if self._ccdm else model_channels
    self.input_blocks = nn.ModuleList([TimestepEmbedSequential(conv_nd(dims, in_channels, first, 3, padding=1))])
    ch = first
    chans = [ch]
    ds = 1
    for level, mult in enumerate(channel_mult):
        for _ in range(num_res_blocks):
            layers = [ResBlock(ch, ted, dropout, out_channels=int(mult * model_channels), dims=dims)]
            ch = int(mult * model_channels)
            if ds in attention_resolutions:
                layers.append(make_attention(ch, num_heads))
            self.input_blocks.append(TimestepEmbedSequential(*layers))
            chans.append(ch)
        if level != len(channel_mult) - 1:
            self.input_blocks.append(TimestepEmbedSequential(Downsample(ch, conv_resample, dims=dims, out_channels=ch)))
            chans.append(ch)
            ds *= 2
    self.middle_block = TimestepEmbedSequential(ResBlock(ch, ted, dropout, dims=dims), make_attention(ch, num_heads),
                                                ResBlock(ch, ted, dropout, dims=dims))
    self.output_blocks = nn.ModuleList([])
    for level, mult in list(enumerate(channel_mult))[::-1]:
        for i in range(num_res_blocks + 1):
            ich = chans.pop()
            layers = [ResBlock(ch + ich, ted, dropout, out_channels=int(model_channels * mult), dims=dims)]
            ch = int(model_channels * mult)
            if ds in attention_resolutions:
                layers.append(make_attention(ch, num_heads_upsample))
            if level and i == num_res_blocks:
                layers.append(Upsample(ch, conv_resample, dims=dims, out_channels=ch))
                ds //= 2
            self.output_blocks.append(TimestepEmbedSequential(*layers))
    return ch


class _UNetBase(nn.Module):
    _ccdm = False

    # ---- time-embedding tables ("timestep-embed epilogue": folded into conv1's per-sample bias) ---------------
    def resblocks(self) -> List[ResBlock]:
        return [m for m in self.modules() if isinstance(m, ResBlock)]

    def time_bias_layout(self, N: int):
        """offsets of every ResBlock's [N, Cout_pad] bias rows inside one flat fp32 row."""
        off, lay = 0, {}
        for rb in self.resblocks():
            cp = pad32(rb.out_channels)
            lay[id(rb)] = (off, cp)
            off += N * cp
        return lay, off

    def time_bias_table(self, timesteps: torch.Tensor, N: int) -> torch.Tensor:
        """timesteps fp32 [S] -> table fp32 [S, total]: for every step the concatenated per-ResBlock biases
        (conv1.bias + emb_layers(time_embed(sinusoid(t)))), identical for the N samples of a step."""
        S = timesteps.shape[0]
        lay, total = self.time_bias_layout(N)
        dev = timesteps.device
        emb = ops.timestep_embedding(timesteps, self.model_channels)
        e1 = ops.linear_f32(emb, f32(self.time_embed[0].weight), f32(self.time_embed[0].bias))
        emb = ops.linear_f32(e1, f32(self.time_embed[2].weight), f32(self.time_embed[2].bias), act_in=True)
        table = torch.zeros((S, total), dtype=torch.float32, device=dev)
        for rb in self.resblocks():
            off, cp = lay[id(rb)]
            tmp = torch.zeros((S, cp), dtype=torch.float32, device=dev)
            rb.time_bias(emb, tmp)
            table[:, off:off + N * cp] = tmp.repeat(1, N)       # plumbing: replicate rows for the batch
        return table

    def time_bias_rows(self, timesteps_per_sample: torch.Tensor) -> torch.Tensor:
        """General per-sample timesteps [N] -> one flat row (eager nn.Module.forward path)."""
        N = timesteps_per_sample.shape[0]
        lay, total = self.time_bias_layout(N)
        dev = timesteps_per_sample.device
        emb = ops.timestep_embedding(timesteps_per_sample.float(), self.model_channels)
        e1 = ops.linear_f32(emb, f32(self.time_embed[0].weight), f32(self.time_embed[0].bias))
        emb = ops.linear_f32(e1, f32(self.time_embed[2].weight), f32(self.time_embed[2].bias), act_in=True)
        row = torch.zeros(total, dtype=torch.float32, device=dev)
        for rb in self.resblocks():
            off, cp = lay[id(rb)]
            rb.time_bias(emb, row[off:off + N * cp].view(N, cp))
        return row

    # ---- channels-last execution ---------------------------------------------------------------------------------
    def forward_cl(self, x: CL, bias_row: torch.Tensor, context: Optional[CL] = None, head_out: Optional[torch.Tensor] = None,
                   head_ddim: Optional[tuple] = None, head_post: Optional[dict] = None) -> CL:
        """x: CL bf16 network input (already concatenated/padded); bias_row: flat fp32 from time_bias_*.
        Returns the head output as fp32 CL [N, D, H, W, pad32(out_channels)] (logits for CCDM, eps for LDM)."""
        N = x.N
        lay, _ = self.time_bias_layout(N)

        def tb(rb):
            off, cp = lay[id(rb)]
            return bias_row[off:off + N * cp]

        ops.stats_begin(x.t.device)        # conv epilogues of this forward leave GroupNorm sums in the (zeroed) arena
        try:
            hs = []
            h = x
            for module in self.input_blocks:
                h = module.run(h, tb, context)
                hs.append(h)
            h = self.middle_block.run(h, tb, context)
            for module in self.output_blocks:
                h = module.run(h, tb, context, skip=hs.pop())
            conv = self.out[2]
            pw, pb = packed_conv(conv, h.Cpad)
            extra = {"ddim": head_ddim} if head_ddim is not None else {}     # DDIM update as the head conv's epilogue (ddim.py:190-204)
            if head_post is not None:
                extra["post"] = head_post                                    # CCDM reverse step as the head conv's epilogue
            return norm_conv(h, self.out[0], True, pw, pb, conv.weight.shape[0], k=_k3(conv.weight), out_f32=True, out=head_out, **extra)
        finally:
            ops.stats_end(x.t.device)


class CCDMUNetModel(_UNetBase):
    """3-D (or 2-D) categorical-diffusion UNet; reference class name `UNetModel` (unet.py:402)."""
    _ccdm = True

    def __init__(self, in_channels, model_channels, out_channels, num_res_blocks, cond_encoded_shape, attention_resolutions,
                 dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None, use_checkpoint=False,
                 use_fp16=False, num_heads=1, num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=False,
                 resblock_updown=False, use_new_attention_order=False, softmax_output=True, ce_head=False,
                 feature_cond_encoder=None, use_spatial_transformer=False, transformer_depth=None, context_dim=None,
                 disabled_sa=False, use_linear_in_transformer=False):
        super().__init__()
        if use_scale_shift_norm or resblock_updown or use_new_attention_order or ce_head or num_classes is not None \
                or use_spatial_transformer:
            raise NotImplementedError("option outside the shipped CCDM configuration (params_eval.yml:58-64)")
        # use_fp16 (unet.py:447,742-756: the reference casts the torso to half precision) is ACCEPTED and has no effect: this engine's
        # torso always runs on bf16 tensors with fp32 accumulation (ops.fp32_validation() is the switch to full precision)
        self.use_fp16 = bool(use_fp16)
        if feature_cond_encoder is not None and feature_cond_encoder.get("type", "none") not in ("none", None):
            raise NotImplementedError("feature_cond_encoder is 'none' in the shipped config; DINO/ResNet features are out of scope")
        if num_heads_upsample == -1:
            num_heads_upsample = num_heads
        self.in_channels, self.model_channels, self.out_channels = in_channels, model_channels, out_channels
        self.num_res_blocks, self.attention_resolutions, self.channel_mult = num_res_blocks, attention_resolutions, channel_mult
        self.dims, self.dtype = dims, torch.float32
        self.num_heads, self.num_head_channels = num_heads, num_head_channels
        self.cond_encoded_shape = cond_encoded_shape
        self.sofmtax_output = softmax_output          # (sic) reference attribute name, unet.py:479
        self.feature_condition_idx = []

        def make_attention(ch, heads):
            return AttentionBlock(ch, num_heads=heads, num_head_channels=num_head_channels)

        ch = _layout(self, dims=dims, in_channels=in_channels, model_channels=model_channels, out_channels=out_channels,
                     num_res_blocks=num_res_blocks, attention_resolutions=attention_resolutions, dropout=dropout,
                     channel_mult=channel_mult, conv_resample=conv_resample, num_heads=num_heads,
                     num_head_channels=num_head_channels, num_heads_upsample=num_heads_upsample, make_attention=make_attention)
        input_ch = int(channel_mult[0] * model_channels)
        head = [normalization(ch), nn.SiLU(), zero_module(conv_nd(dims, input_ch, out_channels, 3, padding=1))]
        if softmax_output:
            head.append(nn.Softmax(dim=1))
        self.out = nn.Sequential(*head)
        self.out_ce = None

    def forward(self, x, input_condition, feature_condition, timesteps, context=None, y=None):
        """Reference signature (unet.py:758). NC[D]HW fp32 in -> {'diffusion_out': NC[D]HW fp32 probs, 'logits': None}."""
        ops.require_gpu(x, "CCDM UNetModel.forward")
        assert y is None, "must specify y if and only if the model is class-conditional"
        cin = x.shape[1] + (input_condition.shape[1] if input_condition is not None else 0)
        xin = ops.to_cl(x, c_pad=pad32(cin))
        if input_condition is not None:
            ops.to_cl(input_condition.to(x.device), out=xin.t, c_offset=x.shape[1], zero_fill=False)
            xin.C = cin
        logits = self.forward_cl(xin, self.time_bias_rows(timesteps.to(x.device).float()))
        out = ops.from_cl(logits, self.dims)
        if self.sofmtax_output:
            out = softmax_dim1(out)
        return {"diffusion_out": out, "logits": None}


def softmax_dim1(logits_nchw: torch.Tensor) -> torch.Tensor:
    """nn.Softmax(dim=1) head (unet.py:715-721) on the HIP sampler kernel's softmax path:
    posterior step at t==1 with x_t-independent identity (a=0, abar=1) reproduces softmax then renormalise."""
    N, K = logits_nchw.shape[:2]
    sp = logits_nchw.shape[2:]
    cl = logits_nchw.permute(0, *range(2, logits_nchw.ndim), 1).contiguous()     # plumbing (view/copy only)
    M = cl.numel() // K
    probs = torch.empty((M, K), dtype=torch.float32, device=cl.device)
    xt = torch.zeros(M, dtype=torch.int32, device=cl.device)
    sc = torch.tensor([0.0, 1.0], dtype=torch.float32, device=cl.device)
    ops.ccdm_posterior_sample(cl.view(M, K), True, xt, sc, K, draw=False, probs_out=probs)
    return probs.view(N, *sp, K).permute(0, logits_nchw.ndim - 1, *range(1, logits_nchw.ndim - 1)).contiguous()


def create_unet_openai(image_size, base_channels, in_channels, out_channels, num_res_blocks, cond_encoded_shape,
                       channel_mult=None, use_checkpoint=False, attention_resolutions=[32, 16, 8], num_heads=1,
                       num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=False, dropout=0, resblock_updown=False,
                       use_fp16=False, use_new_attention_order=False, softmax_output=True, ce_head=False,
                       feature_cond_encoder=None, dims=None):
    """Factory with the reference's defaults (unet_openai/__init__.py:5-66)."""
    if channel_mult is None:
        table = {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4)}
        if image_size not in table:
            raise ValueError(f"unsupported image size: {image_size}")
        channel_mult = table[image_size]
    if dims not in [1, 2, 3]:
        raise NotImplementedError(f"got convnd dims={dims}")
    return CCDMUNetModel(in_channels=in_channels, model_channels=base_channels, out_channels=out_channels,
                         num_res_blocks=num_res_blocks, cond_encoded_shape=cond_encoded_shape,
                         attention_resolutions=attention_resolutions, dropout=dropout, channel_mult=channel_mult, num_classes=None,
                         use_checkpoint=use_checkpoint, use_fp16=use_fp16, num_heads=num_heads, num_head_channels=num_head_channels,
                         num_heads_upsample=num_heads_upsample, use_scale_shift_norm=use_scale_shift_norm,
                         resblock_updown=resblock_updown, use_new_attention_order=use_new_attention_order,
                         softmax_output=softmax_output, ce_head=ce_head, feature_cond_encoder=feature_cond_encoder, dims=dims)


class UNetModel(_UNetBase):
    """LDM UNetModel (openaimodel.py:416-745): optional SpatialTransformer cross-attention."""

    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions, dropout=0,
                 channel_mult=(1, 2, 4, 8), conv_resample=True, dims=3, num_classes=None, use_checkpoint=False, use_fp16=False,
                 num_heads=-1, num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=False, resblock_updown=False,
                 use_new_attention_order=False, use_spatial_transformer=False, transformer_depth=1, context_dim=None,
                 n_embed=None, legacy=True):
        super().__init__()
        if use_spatial_transformer:
            assert context_dim is not None, "use_spatial_transformer requires context_dim"
        if context_dim is not None:
            assert use_spatial_transformer, "context_dim requires use_spatial_transformer"
            context_dim = list(context_dim) if not isinstance(context_dim, int) else context_dim
        if num_heads_upsample == -1:
            num_heads_upsample = num_heads
        if num_heads == -1:
            assert num_head_channels != -1, "Either num_heads or num_head_channels has to be set"
        if num_head_channels == -1:
            assert num_heads != -1, "Either num_heads or num_head_channels has to be set"
        if use_scale_shift_norm or resblock_updown or use_new_attention_order or num_classes is not None or n_embed is not None:
            raise NotImplementedError("option outside the shipped LDM configurations (configs/latent-diffusion/*.yaml)")
        self.image_size, self.in_channels, self.model_channels, self.out_channels = image_size, in_channels, model_channels, out_channels
        self.num_res_blocks, self.attention_resolutions, self.channel_mult = num_res_blocks, attention_resolutions, channel_mult
        self.dims, self.dtype = dims, torch.float32
        self.num_heads, self.num_head_channels, self.num_heads_upsample = num_heads, num_head_channels, num_heads_upsample
        self.use_spatial_transformer, self.context_dim = use_spatial_transformer, context_dim
        self.predict_codebook_ids = False

        def make_attention(ch, heads):
            # head rule incl. legacy=True (openaimodel.py:545-552)
            if num_head_channels == -1:
                nh, dh = heads, ch // heads
            else:
                nh, dh = ch // num_head_channels, num_head_channels
            if legacy:
                dh = ch // nh if use_spatial_transformer else num_head_channels
            if use_spatial_transformer:
                return SpatialTransformer(ch, nh, dh, depth=transformer_depth, context_dim=context_dim)
            return AttentionBlock(ch, num_heads=nh if num_head_channels == -1 else heads, num_head_channels=dh)

        ch = _layout(self, dims=dims, in_channels=in_channels, model_channels=model_channels, out_channels=out_channels,
                     num_res_blocks=num_res_blocks, attention_resolutions=attention_resolutions, dropout=dropout,
                     channel_mult=channel_mult, conv_resample=conv_resample, num_heads=num_heads,
                     num_head_channels=num_head_channels, num_heads_upsample=num_heads_upsample, make_attention=make_attention)
        self.out = nn.Sequential(normalization(ch), nn.SiLU(), zero_module(conv_nd(dims, model_channels, out_channels, 3, padding=1)))

    def context_cl(self, context: Optional[torch.Tensor]) -> Optional[CL]:
        if context is None:
            return None
        N, L, Cc = context.shape
        return ops.to_cl(context.permute(0, 2, 1).contiguous().float(), c_pad=pad32(Cc))   # [N, C, L] -> CL [N,1,1,L,Cpad]

    def forward(self, x, timesteps=None, context=None, y=None, **kwargs):
        """Reference signature (openaimodel.py:713). NCHW fp32 in -> NCHW fp32 eps."""
        ops.require_gpu(x, "UNetModel.forward")
        assert y is None, "must specify y if and only if the model is class-conditional"
        xin = ops.to_cl(x, c_pad=pad32(x.shape[1]))
        eps = self.forward_cl(xin, self.time_bias_rows(timesteps.to(x.device).float()), self.context_cl(context))
        return ops.from_cl(eps, self.dims)
