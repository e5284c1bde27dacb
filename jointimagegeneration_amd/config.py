"""Config plumbing: the reference's two extension points.

1. `instantiate_from_config({"target": "pkg.mod.Class", "params": {...}})` -- the CompVis factory the reference imports
   from its (missing) `models/util.py` (ldm/models/diffusion/ddpm.py:21, sample_diffusion.py:12).  Unmodified configs
   name reference classes by dotted path; TARGET_ALIASES maps those paths onto this package's HIP-backed classes, so
   the shipped yaml files load without edits.
2. CCDM's flat yaml dict with `backbone: "unet_openai"` + `params[params["backbone"]]` kwargs (evaluator.py:215-237).

OmegaConf is not a dependency: yaml.safe_load + `key.sub=value` dot-list overrides give the same behaviour for the
sampling entry points (sample_diffusion.py:519-524).
"""
from __future__ import annotations

import importlib
from typing import Any, Dict, List

import yaml

TARGET_ALIASES = {
    "ldm.modules.diffusionmodules.openaimodel.UNetModel": "jointimagegeneration_amd.unet.UNetModel",
    "ldm.models.autoencoder.AutoencoderKL": "jointimagegeneration_amd.ldm.AutoencoderKL",
    "ldm.models.diffusion.ddpm.LatentDiffusion": "jointimagegeneration_amd.ldm.LatentDiffusion",
    "ldm.modules.encoders.modules.IdentityEncoder": "jointimagegeneration_amd.ldm.IdentityEncoder",
    "ldm.models.diffusion.ddim.DDIMSampler": "jointimagegeneration_amd.ldm.DDIMSampler",
    "ldm.models.diffusion.plms.PLMSSampler": "jointimagegeneration_amd.ldm.PLMSSampler",
    "torch.nn.Identity": "torch.nn.Identity",
}


def get_obj_from_str(string: str):
    string = TARGET_ALIASES.get(string, string)
    module, cls = string.rsplit(".", 1)
    return getattr(importlib.import_module(module), cls)


def instantiate_from_config(config):
    if config in ("__is_first_stage__", "__is_unconditional__"):
        return None
    if "target" not in config:
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**(config.get("params", dict()) or dict()))


def load_yaml(path: str) -> Dict[str, Any]:
    with open(path, "r") as f:
        return yaml.safe_load(f)


def apply_dotlist(cfg: Dict[str, Any], overrides: List[str]) -> Dict[str, Any]:
    """`a.b.c=value` overrides (OmegaConf.from_dotlist semantics for scalars)."""
    for item in overrides:
        key, _, val = item.partition("=")
        node = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = yaml.safe_load(val)
    return cfg


def merge(a: Dict[str, Any], b: Dict[str, Any]) -> Dict[str, Any]:
    out = dict(a)
    for k, v in b.items():
        out[k] = merge(out[k], v) if isinstance(v, dict) and isinstance(out.get(k), dict) else v
    return out
