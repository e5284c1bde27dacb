"""LDM conditional CT sampling entry point: `python -m jointimagegeneration_amd.sample_diffusion -r <logdir|ckpt> -c 50`.

Re-creates the CLI, config and checkpoint surface of latentdiffusion/sample_diffusion.py:165-273,356-433,492-570:
`-r/--resume` (logdir or .ckpt), `-n/--n_samples`, `-e/--eta`, `-c/--custom_steps`, `-l/--logdir`, `--batch_size`, dot-list
overrides; config = logdir/configs/*.yaml merged (+ dot-list); model = instantiate_from_config(config.model) with the
checkpoint's "state_dict" loaded strict=False; sampling under model.ema_scope().  `sample_cond` keeps the reference's
slice loop exactly (Python indexing quirks included) on the reference-shaped API (get_learned_conditioning /
DDIMSampler.sample / decode_first_stage); `GuideGenPipeline.sample_ct` is the all-device fast path of the same loop.
Metrics (LPIPS/FVD), PNG grids and the private datasets are out of scope; the mask comes from `--inputs <dir>` (the
`pred_*.nii.gz` label volumes the stage-1 entry point ddpm_eval writes: the hand-off of README.md:21, one CT volume per mask),
from --mask (.npy label volume [D,H,W]) or is synthetic.
"""
from __future__ import annotations

import argparse
import glob
import os
import sys
import time

import numpy as np
import torch

from .config import apply_dotlist, instantiate_from_config, load_yaml, merge
from . import ops
from .io import load_checkpoint, read_nifti, write_nifti
from .ldm import DDIMSampler, PLMSSampler
from .synth import randomize_parameters, synth_mask_volume


def get_parser():
    p = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    p.add_argument("-r", "--resume", type=str, nargs="?", help="load from logdir or checkpoint in logdir")
    p.add_argument("-n", "--n_samples", type=int, nargs="?", default=1, help="number of samples to draw")
    p.add_argument("-e", "--eta", type=float, nargs="?", default=0.0, help="eta for ddim sampling (0.0 yields deterministic sampling)")
    p.add_argument("-v", "--vanilla_sample", default=False, action="store_true", help="vanilla ancestral sampling (default: DDIM)")
    p.add_argument("--plms", default=False, action="store_true", help="PLMS sampler instead of DDIM")
    p.add_argument("-l", "--logdir", type=str, nargs="?", default="none", help="extra logdir")
    p.add_argument("-c", "--custom_steps", type=int, nargs="?", default=50, help="number of steps for ddim sampling")
    p.add_argument("--batch_size", type=int, nargs="?", default=1)
    p.add_argument("--config", type=str, default=None, help="model yaml when -r is not a log directory")
    p.add_argument("--inputs", type=str, default=None, help="directory of stage-1 label volumes (*.nii.gz / *.nii, as ddpm_eval writes them): "
                   "each is zoomed (order 0) to --slices x --size x --size, rotated and scaled as the reference recipe does, and gets its own CT volume")
    p.add_argument("--mask", type=str, default=None, help=".npy label volume [D,H,W] (labels 0..11); default synthetic ellipsoids")
    p.add_argument("--slices", type=int, default=64)
    p.add_argument("--size", type=int, default=512)
    p.add_argument("--seed", type=int, default=2048)
    return p


def load_model_from_config(config, sd):
    model = instantiate_from_config(config)
    if sd is not None:
        model.load_state_dict(sd, strict=False)
    model.cuda()
    model.eval()
    return model


def load_model(config, ckpt):
    if ckpt and os.path.exists(ckpt):
        pl_sd = load_checkpoint(ckpt)
        global_step = pl_sd.get("global_step", 0)
        model = load_model_from_config(config["model"], pl_sd["state_dict"])
    else:
        print(f"checkpoint {ckpt!r} not found: random-init weights from the seed recipe (synthetic run)", file=sys.stderr)
        model = instantiate_from_config(config["model"])
        randomize_parameters(model, 1024, "ldm.")
        if getattr(model, "use_ema", False):
            model.model_ema.reset_from(model.model)        # the EMA shadow is what ema_scope() samples with
        model.cuda().eval()
        global_step = 0
    return model, global_step


def strip_ckpt_paths(cfg):
    """yaml ckpt_path entries point at the authors' cluster (/mnt/...); sub-model weights come from the main checkpoint."""
    if isinstance(cfg, dict):
        return {k: (None if k == "ckpt_path" and isinstance(v, str) and not os.path.exists(v) else strip_ckpt_paths(v)) for k, v in cfg.items()}
    return cfg


@torch.no_grad()
def stage1_mask_to_wholemask(labels, depth: int, hw: int) -> torch.Tensor:
    """The commented recipe of latentdiffusion/sample_diffusion.py:199-200 on the device: a stage-1 label volume [Dm, Hm, Wm] ->
    `rot90(scipy.ndimage.zoom(mask, (depth, hw, hw) / mask.shape, order=0), k=3, dims=(1, 2)) / 255` as fp32 [depth, hw, hw], through the
    glue kernel the all-device pipeline uses per slice (gg_mask_to_cond_slice: scipy's order-0 index rule in IEEE double)."""
    lab = torch.as_tensor(np.ascontiguousarray(labels)).to(device="cuda", dtype=torch.int32)[None].contiguous()
    vol = torch.empty((depth, hw, hw), dtype=torch.float32, device=lab.device)
    scratch = torch.empty((1, 1, hw, hw, 32), dtype=torch.bfloat16, device=lab.device)
    for d in range(depth):
        ops.mask_to_cond_slice(lab, d, depth, hw, hw, None, scratch, mask_out=vol[d])
    return vol


@torch.no_grad()
def sample_cond(model, instance, n_samples=1, ddim_steps=50, ddim_eta=0.0, noise_seed=None, vanilla=False, plms=False, x_T_tape=None):
    """The reference slice loop (sample_diffusion.py:196-224) on the reference-shaped API. instance["wholemask"] is
    [1, D, H, W, 1] (label/255); returns pred [n, 2, D, H, W] = cat([samples, gen_mask]).  `x_T_tape` (parity runs): one
    [n, C, h, w] start latent per generated slice, in loop order, instead of the generator draw of ddim.py:124."""
    sampler = PLMSSampler(model) if plms else DDIMSampler(model)
    with model.ema_scope():
        wholemask = instance["wholemask"].permute(0, 4, 1, 2, 3).cuda()
        nz = torch.where(wholemask.sum((0, 1, 3, 4)))[0]
        start_layer, end_layer = nz[0], nz[-1]
        shape = (model.channels, model.image_size, model.image_size) if not model.no_first_stage else (1,) + tuple(wholemask.shape[-2:])
        assert wholemask.shape[0] == 1, "batch size should be 1"
        samples = torch.zeros((n_samples,) + wholemask.shape[1:], dtype=torch.float32, device=wholemask.device)
        gen_mask = wholemask.repeat(n_samples, 1, 1, 1, 1)
        g = torch.Generator(device=wholemask.device).manual_seed(noise_seed) if noise_seed is not None else None
        for it, m_ in enumerate(range(start_layer.item() - 1, end_layer.item() + 1)):
            concat_cond = torch.cat([samples[:, :, max(0, m_ - 1)], gen_mask[:, :, m_]], axis=1)
            c = model.get_learned_conditioning(concat_cond)
            if x_T_tape is not None:
                x_T = x_T_tape[it].to(wholemask.device).float()
            else:
                x_T = torch.randn((n_samples,) + shape, generator=g, device=wholemask.device) if g is not None else None
            if vanilla:
                s = model.p_sample_loop(c, (n_samples,) + shape, x_T=x_T, verbose=False)
            else:
                s, _ = sampler.sample(S=ddim_steps, dims=len(shape) - 1, conditioning=c, batch_size=n_samples, shape=shape,
                                      verbose=False, eta=ddim_eta, x_T=x_T)
            ds = model.decode_first_stage(s)
            samples[:, :, m_] = (ds - ds.min()) / (ds.max() - ds.min())
        return torch.cat([samples, gen_mask], dim=1)


def main(argv=None):
    opt, unknown = get_parser().parse_known_args(argv)
    ckpt, logdir = None, opt.logdir
    if opt.resume:
        if os.path.isfile(opt.resume):
            logdir = "/".join(opt.resume.split("/")[:-2]) or "."
            ckpt = opt.resume
        else:
            logdir = opt.resume.rstrip("/")
            ckpt = os.path.join(logdir, "checkpoints", "last.ckpt")
    cfgs = sorted(glob.glob(os.path.join(logdir, "configs", "*.yaml"))) if logdir != "none" else []
    if opt.config:
        cfgs.append(opt.config)
    if not cfgs:
        raise SystemExit("no model config: give -r <logdir with configs/*.yaml> or --config <yaml>")
    config = {}
    for c in cfgs:
        config = merge(config, load_yaml(c))
    config = strip_ckpt_paths(apply_dotlist(config, unknown))
    model, global_step = load_model(config, ckpt)
    print(f"global step: {global_step}", file=sys.stderr)
    out_dir = os.path.join(logdir if logdir != "none" else ".", "samples", f"{global_step:08}")
    os.makedirs(out_dir, exist_ok=True)
    if opt.inputs:
        # stage-1 hand-off (README.md:21): one CT volume per label volume found in the directory, in name order; the start latents of
        # volume i come from the generator seeded with --seed + i
        files = sorted(f for f in glob.glob(os.path.join(opt.inputs, "*.nii*")) if f.endswith((".nii", ".nii.gz")))
        if not files:
            raise SystemExit(f"--inputs {opt.inputs!r}: no *.nii / *.nii.gz label volumes found")
        jobs = [(os.path.basename(f).split(".nii")[0], stage1_mask_to_wholemask(read_nifti(f), opt.slices, opt.size), opt.seed + i) for i, f in enumerate(files)]
    else:
        lab = torch.from_numpy(np.load(opt.mask)).long() if opt.mask else synth_mask_volume(opt.slices, opt.size, opt.size)
        jobs = [("sample", lab.float() / 255.0, opt.seed)]
    for stem, wholemask, seed in jobs:
        instance = {"wholemask": wholemask[None, ..., None]}
        t0 = time.time()
        pred = sample_cond(model, instance, n_samples=opt.n_samples, ddim_steps=opt.custom_steps, ddim_eta=opt.eta, noise_seed=seed,
                           vanilla=opt.vanilla_sample, plms=opt.plms)
        torch.cuda.synchronize()
        for ix, x in enumerate(pred):
            write_nifti(os.path.join(out_dir, f"{stem}_{ix:04d}.nii.gz"), x[0].float().cpu().numpy())
        print(f"sampled {tuple(pred.shape)} in {time.time() - t0:.1f}s -> {out_dir}/{stem}_*.nii.gz", file=sys.stderr)


if __name__ == "__main__":
    main(sys.argv[1:])
