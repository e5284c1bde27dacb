"""Text-conditioning encoder of the CCDM path on the HIP engine.

Mirrors `PreloadedBERTEncoder` (ccdm/ddpm/models/encoder.py:103-123): `depth` BasicTransformerBlocks (self-attention twice
-- attn2 gets no context -- and a GEGLU feed-forward, ccdm/ddpm/models/unet_openai/attention.py:149-170) over CACHED BERT
features `[b, embed_dim, length]`, residual `inputs + outputs`.  Same constructor arguments and state_dict keys, so the
`feature_cond_encoder` entry of an ignite checkpoint (trainer.py:444-463) loads unchanged.  It is built by the reference's
`feature_cond_encoder: {type: "selfattn", embed_dim, n_heads, model_depth, d_head, dropout}` yaml block
(condition_encoder.py:84-99).  The frozen BERT itself (`FrozenBERTEmbedder`, encoder.py:21-100) needs weights that do not
exist offline and is out of scope: the encoder's input is the feature tensor the reference's datasets cache.
The shipped CCDM UNet has no SpatialTransformer and therefore ignores `context`; the LDM UNet with
`use_spatial_transformer` consumes it through the `crossattn` / `hybrid` conditioning keys.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import ops
from .blocks import BasicTransformerBlock
from .ops import CL, pad32


class PreloadedBERTEncoder(nn.Module):
    def __init__(self, embed_dim=768, n_heads=8, depth=4, d_head=64, dropout=0.1):
        super().__init__()
        self.embed_dim = embed_dim
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(embed_dim, n_heads, d_head, dropout=dropout) for _ in range(depth)])

    @torch.no_grad()
    def forward(self, inputs: torch.Tensor) -> torch.Tensor:
        """inputs fp32 [b, embed_dim, length] -> inputs + blocks(inputs), same layout (eval mode: dropout is the identity)."""
        if self.training:
            raise RuntimeError("this engine implements sampling only (training is out of scope, SURVEY.md 2.1 row 5)")
        ops.require_gpu(inputs, "PreloadedBERTEncoder.forward")
        b, c, length = inputs.shape
        assert c == self.embed_dim, f"expected {self.embed_dim} feature channels, got {c}"
        x = ops.to_cl(inputs.float(), c_pad=pad32(c))                    # 'b c l -> b l c' is free in channels-last: CL [b,1,1,l,c]
        for blk in self.transformer_blocks:
            x = blk.run(x, None)
        return inputs + ops.from_cl(x, 1)


def build_feature_cond_encoder(params: dict) -> Optional[nn.Module]:
    """`_build_feature_cond_encoder` (condition_encoder.py:60-108) for the sampling path: 'selfattn' -> PreloadedBERTEncoder,
    'none' -> None; DINO features need network weights and are out of scope."""
    fce = params.get("feature_cond_encoder") or {}
    kind = fce.get("type", "none")
    if kind in ("none", None):
        return None
    if "selfattn" in kind:
        return PreloadedBERTEncoder(fce["embed_dim"], fce["n_heads"], fce["model_depth"], fce["d_head"], fce.get("dropout", 0.0))
    raise NotImplementedError(f"feature_cond_encoder type {kind!r}: DINO / ResNet features are out of scope (need downloaded weights)")
