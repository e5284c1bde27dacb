"""GuideGen volume pipeline on one GPU: CCDM mask (labels) -> stage glue -> autoregressive LDM slices -> CT volume.

Restates the orchestration of ccdm/ddpm/evaluator.py:127-170 (x_T ~ uniform categorical, condition = zeros) and
latentdiffusion/sample_diffusion.py:196-224 (slice loop: cond = [previous generated slice, mask slice] -> cond stage
-> DDIM -> decode -> min-max normalise -> feed back), keeping every tensor on the device in channels-last form.
Volumes are independent units: multi-GPU runs shard volumes over ranks with no collective.

CLI (one process per GPU, under torch.distributed.run or alone):
    python -m jointimagegeneration_amd.pipeline --volumes V --out DIR [--mask-size D H W] [--depth 256] [--hw 512] [--ccdm-steps 250] [--ddim-steps 50]
samples volumes `volume_id mod world_size == rank` and writes `mask_<id>.nii.gz` (uint8 labels) and `ct_<id>.nii.gz` (fp32 in [0, 1])
per volume; volume id v uses seed `--seed + 1000 v` for the mask chain and `+ 1` for the slice loop, whatever the world size.
"""
from __future__ import annotations

import argparse
import os
import sys
import time
from typing import Dict, Optional, Tuple

import torch

from . import ops
from .ccdm import DenoisingModel, DiffusionModel
from .ldm import DDIMSampler, LatentDiffusion
from .ops import CL
from .synth import randomize_parameters
from .unet import create_unet_openai

CCDM_PARAMS = dict(base_channels=64, channel_mult=[1, 2, 2, 4, 5], attention_resolutions=[32, 16, 8], num_heads=1,
                   num_head_channels=32, softmax_output=True)                      # ccdm/params_eval.yml:58-64


def ae_config(in_channels: int, ch: int) -> dict:
    """first/cond stage of configs/latent-diffusion/ruijin-ldm_from_controlnet_ae.yaml:41-94."""
    return dict(target="ldm.models.autoencoder.AutoencoderKL",
                params=dict(embed_dim=4, dims=2, lossconfig=dict(target="torch.nn.Identity"),
                            ddconfig=dict(double_z=True, z_channels=4, resolution=512, in_channels=in_channels, out_ch=in_channels,
                                          ch=ch, ch_mult=[1, 2, 4, 4], num_res_blocks=2, dropout=0.0, dims=2, attn_resolutions=[16, 8])))


LDM_UNET = dict(target="ldm.modules.diffusionmodules.openaimodel.UNetModel",
                params=dict(dims=2, image_size=512, in_channels=8, out_channels=4, model_channels=160, attention_resolutions=[8, 4, 2],
                            num_res_blocks=2, channel_mult=[1, 2, 4, 4, 5], num_head_channels=32))   # …_ae.yaml:17-40


def build_ccdm(K: int = 14, T: int = 250, seed: int = 1024, device="cuda") -> DenoisingModel:
    unet = create_unet_openai(image_size=128, in_channels=K + 1, out_channels=K, num_res_blocks=2, cond_encoded_shape=None, dims=3,
                              **CCDM_PARAMS)
    randomize_parameters(unet, seed, "ccdm.")
    return DenoisingModel(DiffusionModel("cosine", T, K, dims=3), unet, "synthetic", "majority", dims=3).eval().to(device)


def build_ldm(seed: int = 1024, device="cuda", use_ema: bool = False) -> LatentDiffusion:
    m = LatentDiffusion(first_stage_config=ae_config(1, 128), cond_stage_config=ae_config(2, 96), unet_config=LDM_UNET,
                        linear_start=0.0015, linear_end=0.0195, num_timesteps_cond=1, timesteps=1000, first_stage_key="image",
                        cond_stage_key="mask", image_size=64, channels=4, dims=2, use_ema=use_ema)
    randomize_parameters(m, seed, "ldm.")
    return m.eval().to(device)


def _log(msg: str) -> None:
    print(f"[guidegen {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


class GuideGenPipeline:
    progress_every_s = 30.0

    def __init__(self, ccdm: DenoisingModel, ldm: LatentDiffusion, ddim_steps: int = 50):
        self.ccdm, self.ldm, self.ddim_steps = ccdm, ldm, ddim_steps
        self.sampler = DDIMSampler(ldm)
        self.sampler.make_schedule(ddim_steps, ddim_eta=0.0, verbose=False)
        self.stats: Dict[str, float] = {}
        self.use_graph = True
        self._slice_graphs: Dict = {}
        # spatial reduction of the first stage: f = 2^(levels-1) (8 for ch_mult [1,2,4,4], ..._ae.yaml:41-67)
        self.latent_factor = 2 ** (ldm.first_stage_model.decoder.num_resolutions - 1)

    # ---- stage 1 ------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def sample_mask(self, N: int, size: Tuple[int, int, int], seed: int, init_t: Optional[int] = None) -> torch.Tensor:
        """x_T ~ uniform categorical one-hot (evaluator.py:135-136), condition = zeros; returns int32 labels [N,D,H,W]."""
        dev = self.ccdm.diffusion.betas.device
        K = self.ccdm.diffusion.num_classes
        g = torch.Generator(device=dev).manual_seed(seed)
        x_T = torch.randint(0, K, (N,) + tuple(size), generator=g, device=dev, dtype=torch.int32)
        self.ccdm.philox_seed = seed
        cond = torch.zeros((N, 1) + tuple(size), device=dev)
        labels, _ = self.ccdm.sample_labels(x_T, cond, init_t)
        return labels

    def _slice_engine(self, N: int, hw: int, dev, st) -> Dict:
        """Static buffers + hipGraphs of the two per-slice networks (cond-stage encode, first-stage decode) for one shape."""
        ldm = self.ldm
        lat, Cz = hw // self.latent_factor, ldm.channels
        key = (N, hw, str(dev))
        token = (id(st), ops.weights_token(ldm.cond_stage_model), ops.weights_token(ldm.first_stage_model))
        sg = self._slice_graphs.get(key)
        if sg is not None and sg["token"] == token:
            return sg
        sg = dict(token=token, cond_in=torch.empty((N, 1, hw, hw, 32), dtype=torch.bfloat16, device=dev),
                  z=torch.zeros((N, 1, lat, lat, 32), dtype=torch.bfloat16, device=dev),
                  ds=torch.empty((N, hw, hw), dtype=torch.float32, device=dev), enc=None, dec=None, slice=None, mom=None, warmed=False)
        cond_in, zbuf = sg["cond_in"], sg["z"]

        def encode():
            sg["mom"] = ldm.cond_stage_model.encode_moments_cl(CL(cond_in, 2))         # fp32 CL [N,1,lat,lat,32]; mode() = mean
            st["unet_in"][..., Cz:2 * Cz].copy_(sg["mom"].t[..., :Cz])                # c_concat = posterior mean (plumbing copy)

        def decode():
            zbuf[..., :Cz].copy_(st["x"] * (1.0 / ldm.scale_factor))
            dec = ldm.first_stage_model.decode_cl(CL(zbuf, Cz))                        # fp32 CL [N,1,hw,hw,32]
            sg["ds"].copy_(dec.t[..., 0].reshape(N, hw, hw))
            ops.minmax_normalise(sg["ds"], out=sg["ds"])

        sg["encode"], sg["decode"] = encode, decode
        self._slice_graphs[key] = sg
        return sg

    @torch.no_grad()
    def time_slice_stages(self, N: int = 1, hw: int = 512) -> Dict[str, float]:
        """HIP-event times (ms) of the three per-slice stages on the engine's own buffers, as the slice loop runs them
        (captured graphs when enabled): cond-encode, one DDIM step (average over the S steps of a slice), decode."""
        dev = self.ldm.device
        lat, Cz = hw // self.latent_factor, self.ldm.channels
        st = self.sampler.prepare_state(N, Cz, (lat, lat), dev, Cz)
        sg = self._slice_engine(N, hw, dev, st)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]

        sg["cond_in"].zero_()
        st["x"].normal_()
        st["unet_in"][..., :Cz].copy_(st["x"])
        if self.use_graph:                                    # per-stage graphs for the timing only (the slice loop replays ONE graph)
            if not sg["warmed"]:
                sg["encode"](); sg["decode"]()
            if sg["enc"] is None:
                sg["enc"], sg["dec"] = ops.capture_graph(sg["encode"]), ops.capture_graph(sg["decode"])

        def run(g, fn):
            g.replay() if (self.use_graph and g is not None) else fn()

        for rep in range(3):                                  # eager warm-up, chain-graph capture + first replay, timed
            ev[0].record()
            run(sg["enc"], sg["encode"])
            ev[1].record()
            self.sampler.run_steps(st, None, 0.0, None)
            ev[2].record()
            run(sg["dec"], sg["decode"])
            ev[3].record()
        torch.cuda.synchronize()
        return {"encode_ms": ev[0].elapsed_time(ev[1]), "ddim_step_ms": ev[1].elapsed_time(ev[2]) / self.ddim_steps,
                "decode_ms": ev[2].elapsed_time(ev[3])}

    # ---- stage 2 ------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def sample_ct(self, labels: torch.Tensor, depth: int, hw: int, seed: int, max_slices: Optional[int] = None,
                  x_T_tape=None) -> torch.Tensor:
        """labels int32 [N,Dm,Hm,Wm] -> CT volume fp32 [N, depth, hw, hw] in [0,1]; slice m conditioned on slice m-1.
        `x_T_tape` (parity runs): one [N, Cz, lat, lat] start latent per generated slice, in loop order, instead of the seeded draw."""
        ldm, sampler, S = self.ldm, self.sampler, self.ddim_steps
        dev = labels.device
        N = labels.shape[0]
        lat = hw // self.latent_factor
        Cz = ldm.channels
        g = torch.Generator(device=dev).manual_seed(seed)
        samples = torch.zeros((depth, N, hw, hw), dtype=torch.float32, device=dev)        # slice-major: one slice is contiguous
        # slices whose upsampled mask is non-empty (sample_diffusion.py:202); python indexing of the reference loop kept
        Dm = labels.shape[1]
        nz = (labels != 0).flatten(2).any(-1).any(0)                                       # [Dm]
        idx = torch.nonzero(nz[ops.zoom0_index(Dm, depth).to(dev)]).flatten()       # same order-0 rule as the glue kernel
        start, end = (int(idx[0]), int(idx[-1])) if idx.numel() else (1, 0)
        st = sampler.prepare_state(N, Cz, (lat, lat), dev, Cz)
        todo = list(range(start - 1, end + 1))
        if max_slices is not None:
            todo = todo[:max_slices]
        sg = self._slice_engine(N, hw, dev, st)
        cond_in, encode, decode = sg["cond_in"], sg["encode"], sg["decode"]

        t_last = time.time()
        for it, m in enumerate(todo):
            if time.time() - t_last > self.progress_every_s:
                _log(f"LDM slice {it}/{len(todo)}")
                t_last = time.time()
            mm = m % depth
            prev = samples[max(0, m - 1) % depth]
            ops.mask_to_cond_slice(labels, mm, depth, hw, hw, prev, cond_in)
            # drawn in the reference's NCHW element order (ddim.py:124 `torch.randn(shape)`), so that a seeded generator gives
            # sample_diffusion.sample_cond and this loop the same x_T; stored channels-last (plumbing copies)
            x_T = x_T_tape[it].to(dev).float() if x_T_tape is not None else torch.randn((N, Cz, lat, lat), generator=g, device=dev)
            x_T = x_T.permute(0, 2, 3, 1).reshape(N, 1, lat, lat, Cz)
            st["x"].copy_(x_T)
            st["unet_in"][..., :Cz].copy_(x_T)                                         # fp32 -> bf16
            if not (self.use_graph and sampler.chain_graphable(st)):
                encode()
                sampler.run_steps(st, None, 0.0, None)
                decode()
            elif not sg["warmed"]:
                encode()                                                               # eager once: fills the repack caches
                sampler.chain(st)
                decode()
                sg["warmed"] = True
            else:
                # ONE hipGraph per slice: cond-encode, the S DDIM steps (each reading its own rows of the bias / scalar tables) and the
                # decode; the host's share of a slice is the glue kernel, the x_T draw and this replay
                if sg["slice"] is None:
                    sg["slice"] = ops.capture_graph(lambda: (encode(), sampler.chain(st), decode()))
                sg["slice"].replay()
            samples[mm].copy_(sg["ds"])
        return samples.permute(1, 0, 2, 3)

    @torch.no_grad()
    def run_volume(self, N: int = 1, mask_size=(128, 128, 128), depth: int = 256, hw: int = 512, seed: int = 1024,
                   ccdm_init_t: Optional[int] = None, max_slices: Optional[int] = None):
        t0 = time.time()
        labels = self.sample_mask(N, mask_size, seed, ccdm_init_t)
        torch.cuda.synchronize()
        t1 = time.time()
        ct = self.sample_ct(labels, depth, hw, seed + 1, max_slices)
        torch.cuda.synchronize()
        self.stats = {"ccdm_s": t1 - t0, "ldm_s": time.time() - t1}
        _log(f"volume done: CCDM {t1 - t0:.1f}s, LDM {time.time() - t1:.1f}s")
        return labels, ct


def main(argv=None) -> None:
    """Sharded full-pipeline entry point (volume_id -> rank = volume_id mod world_size, no collective on the data path)."""
    from . import distributed as ggd
    from .io import write_nifti
    ap = argparse.ArgumentParser(description="GuideGen volumes: CCDM mask -> LDM CT, sharded over the ranks")
    ap.add_argument("--volumes", type=int, required=True, help="number of volumes of the whole job")
    ap.add_argument("--out", default="guidegen_out")
    ap.add_argument("--mask-size", type=int, nargs=3, default=(128, 128, 128))
    ap.add_argument("--depth", type=int, default=256)
    ap.add_argument("--hw", type=int, default=512)
    ap.add_argument("--classes", type=int, default=14)
    ap.add_argument("--ccdm-steps", type=int, default=250)
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--seed", type=int, default=1024)
    ap.add_argument("--max-slices", type=int, default=None, help="DEV ONLY: truncate the slice loop")
    args = ap.parse_args(argv)
    rank, local, world = ggd.env_rank_world()
    dry = os.environ.get("GG_PIPELINE_DRY") == "1"          # CPU rehearsal of the sharding (tests): no model, no GPU
    os.makedirs(args.out, exist_ok=True)
    mine = ggd.shard(args.volumes, rank, world)
    if dry:
        ggd.init("gloo")
        pipe = None
    else:
        assert torch.cuda.is_available(), "the GuideGen engine needs an MI355X (no CPU fallback)"
        dev = torch.device("cuda", local)
        torch.cuda.set_device(dev)
        ggd.init("nccl", dev)
        pipe = GuideGenPipeline(build_ccdm(args.classes, args.ccdm_steps, 1024, dev), build_ldm(1024, dev), ddim_steps=args.ddim_steps)
    t0 = time.time()
    for vid in mine:
        seed = args.seed + 1000 * vid
        if dry:
            with open(os.path.join(args.out, f"ct_{vid:04d}.txt"), "w") as f:
                f.write(f"rank {rank} world {world} seed {seed}\n")
            continue
        labels, ct = pipe.run_volume(N=1, mask_size=tuple(args.mask_size), depth=args.depth, hw=args.hw, seed=seed, max_slices=args.max_slices)
        write_nifti(os.path.join(args.out, f"mask_{vid:04d}.nii.gz"), labels[0].to(torch.uint8).cpu().numpy())
        write_nifti(os.path.join(args.out, f"ct_{vid:04d}.nii.gz"), ct[0].float().cpu().numpy())
    print(f"[rank {rank}/{world}] volumes {mine} in {time.time() - t0:.1f}s -> {args.out}", file=sys.stderr)
    ggd.finalize()


if __name__ == "__main__":
    main(sys.argv[1:])
