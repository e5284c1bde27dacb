"""CCDM mask sampling entry point: `python -m jointimagegeneration_amd.ddpm_eval params_eval.yml [exp_name]`.

Re-creates the CLI/config/checkpoint surface of ccdm/ddpm_eval.py:16-57 + ccdm/ddpm/evaluator.py:127-170,215-237,326-393
without ignite, datasets or metrics (out of scope, SURVEY.md 2.1 rows 4,7): seeds, flat yaml dict,
`build_model(..., backbone, params[params["backbone"]], ...)`, ignite-style checkpoint {"model", "average_model"} holding the
UNet's state_dict, x_T ~ uniform one-hot, condition image = zeros, output label = argmax.  Inputs are synthetic (the
hospital dataset is private): the volume extent comes from --size or the yaml key `input_size`.
"""
from __future__ import annotations

import argparse
import os
import random
import sys
import time

import numpy as np
import torch
import yaml

from . import distributed as ggd
from .ccdm import build_model
from .encoder import build_feature_cond_encoder
from .io import load_checkpoint, write_nifti
from .synth import randomize_parameters


def set_seeds(seed: int):
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed % 2 ** 32)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)


def build_from_params(params: dict, size, num_classes: int):
    input_shapes = [(1,) + tuple(size), (num_classes,) + tuple(size)]
    return build_model(time_steps=params["time_steps"], schedule=params["beta_schedule"], schedule_params=params.get("beta_schedule_params"),
                       input_shapes=input_shapes, cond_encoded_shape=None, backbone=params["backbone"],
                       backbone_params=params[params["backbone"]], dataset_file=params.get("dataset_file", "synthetic"),
                       step_T_sample=params.get("evaluation_vote_strategy"), feature_cond_encoder=params.get("feature_cond_encoder"),
                       dims=params.get("dims", 3))


def load_weights(model, params: dict, log=print) -> str:
    path = params.get("load_from")
    if path and os.path.exists(path):
        ckpt = load_checkpoint(path)
        sd = ckpt.get("average_model", ckpt.get("model", ckpt))        # Polyak average is what evaluator.predict uses
        missing, unexpected = model.unet.load_state_dict(sd, strict=False)
        log(f"loaded {path}: {len(missing)} missing / {len(unexpected)} unexpected keys")
        return path
    log(f"checkpoint {path!r} not found: using random-init weights from the seed recipe (synthetic run)")
    randomize_parameters(model.unet, 1024, "ccdm.")
    return "random-init"


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    ap.add_argument("params_file", nargs="?", default="params_eval.yml")
    ap.add_argument("exp_name", nargs="?", default="local_test")
    ap.add_argument("--size", type=int, nargs=3, default=None, help="D H W of the mask volume (default: yaml input_size or 64 128 128)")
    ap.add_argument("--num-classes", type=int, default=None)
    ap.add_argument("--num-volumes", type=int, default=None, help="volumes to sample (default: yaml batch_size, reference forces 2)")
    ap.add_argument("--steps", type=int, default=None, help="run K evenly spaced reverse steps (reference convention t = 10000+K)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args(argv)
    set_seeds(1024)
    with open(args.params_file, "r") as f:
        params = yaml.safe_load(f)
    params["batch_size"] = 2 if args.num_volumes is None else args.num_volumes        # ddpm_eval.py:51
    size = tuple(args.size or params.get("input_size") or (64, 128, 128))
    K = args.num_classes or params.get("num_classes", 12)
    rank, local, world = ggd.env_rank_world()
    assert torch.cuda.is_available(), "the GuideGen engine needs an MI355X (no CPU fallback)"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    ggd.init("nccl", dev)
    model = build_from_params(params, size, K).eval()
    load_weights(model, params, log=lambda m: print(f"[rank {rank}] {m}", file=sys.stderr))
    model = model.to(dev)
    # text conditioning (evaluator.py:166-169): the 'selfattn' feature encoder runs over the cached BERT features of the volume;
    # synthetic features here (the hospital reports are private).  The shipped UNet has no SpatialTransformer and ignores the
    # resulting context, exactly as in the reference.
    fce = build_feature_cond_encoder(params)
    if fce is not None:
        path = params.get("load_from")
        sd_fce = None
        if path and os.path.exists(path):
            ck = load_checkpoint(path)
            sd_fce = ck.get("average_feature_cond_encoder", ck.get("feature_cond_encoder"))
        try:
            if sd_fce is not None:      # a DDP / DataParallel-wrapped encoder saves its keys as `module.*` (condition_encoder.py:91-97)
                fce.load_state_dict({(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd_fce.items()})
            else:
                randomize_parameters(fce, 1024, "fce.")
            fce = fce.eval().to(dev)
            feats = torch.randn((1, fce.embed_dim, int(params.get("context_length", 512))), generator=torch.Generator(device=dev).manual_seed(7), device=dev)
            context = fce(feats)        # computed as evaluator.py:166-169 does; the shipped UNet takes no context and ignores it
            print(f"[rank {rank}] feature_cond_encoder context {tuple(context.shape)} (not consumed by the shipped UNet)", file=sys.stderr)
        except RuntimeError as e:       # load_state_dict mismatch (the reference's load would raise, ccdm/ddpm/trainer.py:444-463)
            if getattr(model.unet, "context_dim", None) is not None:
                raise                   # a UNet that consumes the context must not silently run unconditioned
            # the shipped UNet takes no context: the eval does not need the encoder, say so and go on
            print(f"[rank {rank}] WARNING: feature_cond_encoder skipped ({type(e).__name__}: {e})", file=sys.stderr)
    out_dir = args.out or os.path.join(params.get("output_path", "."), args.exp_name)
    os.makedirs(out_dir, exist_ok=True)
    init_t = None if args.steps is None else 10000 + args.steps
    t0 = time.time()
    for vid in ggd.shard(params["batch_size"], rank, world):                          # volumes are independent units
        g = torch.Generator(device=dev).manual_seed(1024 + vid)
        x_T = torch.randint(0, K, (1,) + size, generator=g, device=dev, dtype=torch.int32)   # uniform categorical x_T
        model.philox_seed = 1024 + vid
        labels, _ = model.sample_labels(x_T, torch.zeros((1, 1) + size, device=dev), init_t)
        write_nifti(os.path.join(out_dir, f"pred_{vid:04d}.nii.gz"), labels[0].to(torch.uint8).cpu().numpy())
    torch.cuda.synchronize()
    print(f"[rank {rank}] sampled {len(ggd.shard(params['batch_size'], rank, world))} volume(s) of {size} in {time.time() - t0:.1f}s -> {out_dir}",
          file=sys.stderr)
    ggd.finalize()


if __name__ == "__main__":
    main(sys.argv[1:])
