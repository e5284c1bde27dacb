"""Deterministic synthetic weights and inputs (there are no checkpoints or data offline).

Every floating-point parameter is drawn from its own CPU generator seeded by
crc32(name) ^ seed, so the values depend only on (name, shape, seed) -- not on
constructor order -- and can be regenerated identically for the reference
modules, the CPU oracle and the HIP engine ("weights are
regenerated from the seed recipe on both sides").
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Iterable, Tuple

import torch


def _gen(name: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def synth_tensor(name: str, shape: Tuple[int, ...], seed: int) -> torch.Tensor:
    """fan-in scaled normal for matrices/conv kernels, 1+0.1n for norm scales, 0.05n for biases."""
    g = _gen(name, seed)
    shape = tuple(shape)
    if len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        return torch.randn(shape, generator=g) * (1.0 / math.sqrt(fan_in))
    if name.endswith("weight"):
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    return 0.05 * torch.randn(shape, generator=g)


@torch.no_grad()
def randomize_parameters(module: torch.nn.Module, seed: int = 1024, prefix: str = "") -> None:
    """In-place: overwrite every parameter of `module` (incl. zero_module convs) with synth_tensor."""
    for name, p in module.named_parameters():
        p.copy_(synth_tensor(prefix + name, tuple(p.shape), seed).to(p.dtype))


def synth_state_dict(named_shapes: Iterable[Tuple[str, Tuple[int, ...]]], seed: int = 1024) -> Dict[str, torch.Tensor]:
    return {n: synth_tensor(n, tuple(s), seed) for n, s in named_shapes}


def synth_mask_volume(D: int, H: int, W: int, n_labels: int = 12) -> torch.Tensor:
    """Nested-ellipsoid label volume with labels 0..n_labels-1, int64 [D,H,W]."""
    z = torch.linspace(-1, 1, D)[:, None, None]
    y = torch.linspace(-1, 1, H)[None, :, None]
    x = torch.linspace(-1, 1, W)[None, None, :]
    r = torch.sqrt((z / 0.9) ** 2 + (y / 0.8) ** 2 + (x / 0.7) ** 2)
    lab = torch.clamp(((1.0 - r) * n_labels).floor(), min=0, max=n_labels - 1)
    return lab.to(torch.int64)
