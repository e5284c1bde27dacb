"""CCDM volumetric mask sampler on the HIP engine.

Mirrors (names, signatures, buffers) ccdm/ddpm/models/diffusion_denoising.py:18-227, builder.py:14-53 and the
input conventions of ccdm/ddpm/evaluator.py:127-170.  The reverse chain keeps its state as int32 labels on the
device; one denoising step = UNet forward (channels-last bf16, MFMA) + ONE fused per-voxel kernel
(softmax -> theta_post_prob -> clamp -> renormalise -> exponential-race sample -> one-hot write-back), and the
fixed-shape step is captured in a hipGraph (torch.cuda.CUDAGraph) and replayed with per-step tables.
"""
from __future__ import annotations

import math
from typing import Any, Dict, List, Optional, Sequence, Tuple, cast

import numpy as np
import torch
from torch import Tensor, nn

from . import ops
from .ops import CL, pad32
from .unet import create_unet_openai

__all__ = ["DiffusionModel", "DenoisingModel", "build_model", "linear_schedule", "cosine_schedule"]


def linear_schedule(time_steps: int, start=1e-2, end=0.2):
    betas = torch.linspace(start, end, time_steps)
    alphas = 1 - betas
    return betas, alphas, torch.cumprod(alphas, dim=0)


def cosine_schedule(time_steps: int, s: float = 8e-3):
    """Reproduces the reference values bit-for-bit, quirks included (diffusion_denoising.py:25-39):
    `s` is overridden to 0.008, cumalphas is sampled at t=0..T-1 (not a cumprod), betas are capped at 0.999."""
    s = 0.008
    t = torch.arange(0, time_steps)
    cumalphas = torch.cos(((t / time_steps + s) / (1 + s)) * (math.pi / 2)) ** 2
    f = lambda u: math.cos((u + s) / (1.0 + s) * math.pi / 2) ** 2
    betas = torch.tensor([min(1 - f((i + 1) / time_steps) / f(i / time_steps), 0.999) for i in range(time_steps)])
    return betas, 1 - betas, cumalphas


class DiffusionModel(nn.Module):
    """Schedule buffers `betas/alphas/cumalphas` (diffusion_denoising.py:42-71)."""
    betas: Tensor
    alphas: Tensor
    cumalphas: Tensor

    def __init__(self, schedule: str, time_steps: int, num_classes: int, schedule_params=None, dims=3):
        super().__init__()
        fn = {"linear": linear_schedule, "cosine": cosine_schedule}[schedule]
        betas, alphas, cumalphas = fn(time_steps, **schedule_params) if schedule_params is not None else fn(time_steps)
        self.dims = dims
        self.register_buffer("betas", betas)
        self.register_buffer("alphas", alphas)
        self.register_buffer("cumalphas", cumalphas)
        self.num_classes = num_classes

    @property
    def time_steps(self):
        return len(self.betas)

    def step_scalars(self, t_values: Sequence[int]) -> Tensor:
        """fp32 [S, 2] = (alphas[t-1], cumalphas[t-2]) with the t==1 overrides (a=0, abar=1) of
        theta_post_prob (diffusion_denoising.py:114-122)."""
        al, ca = self.alphas.detach().cpu(), self.cumalphas.detach().cpu()
        rows = [(0.0, 1.0) if t == 1 else (float(al[t - 1]), float(ca[t - 2])) for t in t_values]
        return torch.tensor(rows, dtype=torch.float32)

    def theta_post_prob(self, xt: Tensor, theta_x0: Tensor, t: Tensor) -> Tensor:
        """Reference signature (NC[D]HW one-hot xt, probs theta_x0, 1-based t [B]); evaluated by the fused HIP kernel
        (one launch per distinct t). Returns the UN-clamped-equivalent normalised posterior (values >= 1e-12)."""
        ops.require_gpu(xt, "theta_post_prob")
        K = self.num_classes
        nd = xt.ndim
        perm = (0,) + tuple(range(2, nd)) + (1,)
        inv = (0, nd - 1) + tuple(range(1, nd - 1))
        out = torch.empty_like(theta_x0.permute(perm).contiguous())
        for b in range(xt.shape[0]):
            sc = self.step_scalars([int(t[b])]).to(xt.device)[0]
            p0 = theta_x0[b:b + 1].permute(perm).contiguous().float()
            lab = xt[b:b + 1].argmax(dim=1).to(torch.int32).contiguous().view(-1)
            M = lab.numel()
            ops.ccdm_posterior_sample(p0.view(M, K), False, lab, sc, K, draw=False, probs_out=out[b].view(M, K))
        return out.permute(inv)


class DenoisingModel(nn.Module):
    """`forward(x, condition, ...)` in eval mode runs the whole reverse chain (diffusion_denoising.py:142-227)."""

    def __init__(self, diffusion: DiffusionModel, unet: nn.Module, dataset_file: str, step_T_sample: str = "majority", dims=3):
        super().__init__()
        self.diffusion, self.unet, self.dims = diffusion, unet, dims
        self.dataset_file, self.step_T_sample = dataset_file, step_T_sample
        self.philox_seed = 1024           # RNG of the throughput path (counter-based, in-kernel)
        self.use_graph = True
        self._graph_cache: Dict[Any, Any] = {}

    @property
    def time_steps(self):
        return self.diffusion.time_steps

    def forward(self, x: Tensor, condition: Tensor, feature_condition: Tensor = None, t: Optional[Tensor] = None,
                label_ref_logits: Optional[Tensor] = None, validation: bool = False, context=None, rng_tapes=None) -> dict:
        if self.training:
            raise RuntimeError("this engine implements sampling only (training is out of scope, SURVEY.md 2.1 row 5)")
        if validation:
            return self.forward_step(x, condition, feature_condition, t, context=context)
        init_t = None if t is None else cast(int, int(t.item()))
        return self.forward_denoising(x, condition, feature_condition, init_t, label_ref_logits, context=context, rng_tapes=rng_tapes)

    def forward_step(self, x, condition, feature_condition, t, context=None):
        return self.unet(x, condition, feature_condition=feature_condition, timesteps=t, context=context)

    # -------------------------------------------------------------------------------------------------
    def t_values(self, init_t: Optional[int]) -> List[int]:
        T = self.time_steps
        if init_t is None:
            init_t = T
        if init_t > 10000:                      # "t = 10000+K" sub-sampling (diffusion_denoising.py:190-197)
            K = init_t % 10000
            assert 0 < K <= T
            return list(range(K, 0, -1)) if K == T else [round(v) for v in np.linspace(T, 1, K)]
        return list(range(init_t, 0, -1))

    @torch.no_grad()
    def forward_denoising(self, x: Tensor, condition: Tensor, feature_condition: Tensor = None, init_t: Optional[int] = None,
                          label_ref_logits: Optional[Tensor] = None, context: Tensor = None, rng_tapes=None) -> dict:
        if label_ref_logits is not None:
            raise NotImplementedError("label_ref_logits guidance is dead code in the reference (undefined guidance_fn)")
        if feature_condition is not None:
            raise NotImplementedError("feature_condition is always None on the shipped path (evaluator.py:169)")
        ops.require_gpu(x, "DenoisingModel.forward_denoising")
        labels = x.argmax(dim=1).to(torch.int32).contiguous()                   # x_T one-hot -> labels (plumbing)
        lab, probs = self.sample_labels(labels, condition, init_t, rng_tapes)
        K = self.diffusion.num_classes
        nd = x.ndim
        inv = (0, nd - 1) + tuple(range(1, nd - 1))
        if self.step_T_sample is None or self.step_T_sample == "majority":
            out = torch.nn.functional.one_hot(lab.long(), K).permute(inv)       # int64 one-hot like max_prob_sample
        elif self.step_T_sample == "confidence":
            out = probs.view(*lab.shape, K).permute(inv)
        else:
            raise ValueError(f"step_T_sample={self.step_T_sample}")
        return {"diffusion_out": out}

    @torch.no_grad()
    def sample_labels(self, labels: Tensor, condition: Optional[Tensor], init_t: Optional[int] = None, rng_tapes=None,
                      trace: Optional[list] = None) -> Tuple[Tensor, Tensor]:
        """labels int32 [N, (D,) H, W] on the GPU -> (final labels int32, final normalised posterior fp32 [M, K])."""
        dev = labels.device
        K = self.diffusion.num_classes
        unet = self.unet
        N = labels.shape[0]
        sp = tuple(labels.shape[1:])
        sp3 = (1,) * (3 - len(sp)) + sp
        M = labels.numel()
        tv = self.t_values(init_t)
        S = len(tv)
        scal = self.diffusion.step_scalars(tv).to(dev)
        table = unet.time_bias_table(torch.tensor(tv, dtype=torch.float32, device=dev), N)
        cin = unet.in_channels
        xin = torch.zeros((N,) + sp3 + (pad32(cin),), dtype=torch.float32 if ops.FP32 else torch.bfloat16, device=dev)
        lab = labels.contiguous().view(-1).clone()
        ops.labels_to_onehot(lab, K, xin.view(M, -1))                 # channels [0,K) one-hot, rest zero
        if condition is not None:                                     # unet.py:774-775: cat([x, input_condition], 1)
            ops.to_cl(condition.to(dev), out=xin, c_offset=K, zero_fill=False)
        logits = torch.empty((N,) + sp3 + (pad32(K),), dtype=torch.float32, device=dev)
        probs = torch.empty((M, K), dtype=torch.float32, device=dev)
        cur_bias = torch.empty_like(table[0])
        cur_scal = torch.empty(2, dtype=torch.float32, device=dev)
        cur_off = torch.zeros(1, dtype=torch.int64, device=dev)
        xcl = CL(xin, cin)

        def step(draw: bool, E=None, want_probs=False):
            bf16_in = xin.dtype == torch.bfloat16
            # softmax + posterior + draw as the head conv's epilogue where the kernel can (ops.conv `post`): the fp32 logits (128 B per
            # voxel) are then never written; steps that return the posterior itself (the last one, traces) take the sampler kernel
            post = dict(xt=lab, scalars=cur_scal, K=K, E=E, philox_seed=self.philox_seed, philox_offset=cur_off, draw=draw, labels_out=lab,
                        onehot_out=xin.view(M, -1)) if (bf16_in and not want_probs) else None
            head = unet.forward_cl(xcl, cur_bias, head_out=logits, head_post=post)
            self.last_step_fused = head.fused_post
            if head.fused_post:
                return
            ops.ccdm_posterior_sample(logits.view(M, -1), True, lab, cur_scal, K, E=E, philox_seed=self.philox_seed,
                                      philox_offset=cur_off, draw=draw, labels_out=lab,
                                      probs_out=probs if want_probs else None, onehot_out=xin.view(M, -1) if bf16_in else None)
            if not bf16_in:                                           # fp32 validation mode: the one-hot input is refreshed by an index scatter
                ops.labels_to_onehot(lab, K, xin.view(M, -1))

        if rng_tapes is not None:
            need = sum(1 for t in tv if t > 1)                        # one exponential tape per DRAWING step (t = 1 takes the argmax)
            if len(rng_tapes) < need:
                raise ValueError(f"sample_labels: {need} drawing steps (t = {tv[0]} .. 2) need {need} rng tapes, got {len(rng_tapes)}")
        use_graph = self.use_graph and rng_tapes is None and trace is None and S > 3 and not ops.FP32
        graph, warmed, ti = None, False, 0
        for i, t in enumerate(tv):
            cur_bias.copy_(table[i]); cur_scal.copy_(scal[i]); cur_off.fill_(t)
            last, draw = (i == S - 1), (t > 1)
            if use_graph and draw and not last:
                if not warmed:
                    step(True)                                   # eager warm-up: fills the weight-repack cache
                    warmed = True
                else:
                    if graph is None:                            # capture the fixed-shape step once (hipGraph)
                        graph = ops.capture_graph(lambda: step(True))
                    graph.replay()
            else:
                E = None
                if rng_tapes is not None and draw:
                    E = rng_tapes[ti].to(dev).contiguous(); ti += 1
                step(draw, E, want_probs=last or trace is not None)
            if trace is not None:
                trace.append(dict(t=t, labels=lab.clone().view(labels.shape), probs=probs.clone()))
        return lab.view(labels.shape), probs


def build_model(time_steps: int, schedule: str, schedule_params, input_shapes, cond_encoded_shape, backbone: str,
                backbone_params: Dict[str, Any], dataset_file: str, step_T_sample: str = None, feature_cond_encoder: dict = None,
                dims: int = 3) -> DenoisingModel:
    """Same contract as ccdm/ddpm/models/builder.py:14-53 (in_channels = num_classes + img_channels)."""
    img_shape, label_shape, *_ = input_shapes
    img_channels, num_classes = img_shape[0], label_shape[0]
    diffusion = DiffusionModel(schedule, time_steps, num_classes, schedule_params=schedule_params, dims=dims)
    if backbone != "unet_openai":
        raise NotImplementedError(f"backbone {backbone}")
    fce = feature_cond_encoder if (feature_cond_encoder or {}).get("type", "none") not in ("none", None) else None
    model = create_unet_openai(image_size=min(img_shape[1], img_shape[2]), in_channels=num_classes + img_channels,
                               out_channels=num_classes, num_res_blocks=2, cond_encoded_shape=cond_encoded_shape,
                               feature_cond_encoder=fce, dims=dims, **backbone_params)
    return DenoisingModel(diffusion, model, dataset_file, step_T_sample, dims)
