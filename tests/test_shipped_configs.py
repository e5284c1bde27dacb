"""CPU suite, BUILD CONTAINER ONLY (skipped where /root/reference is absent, e.g. on the GPU box): the reference's shipped config
files load UNEDITED through this package's config plumbing -- `ccdm/params_eval.yml` via `ddpm_eval.build_from_params`
(evaluator.py:215-237) and every `latentdiffusion/configs/**/*.yaml` via `config.instantiate_from_config` with the reference's own
dotted `target:` paths (TARGET_ALIASES) -- and the modules they build expose the reference's state_dict surfaces
(tests/golden/surfaces_full.json, captured from the reference constructors)."""
import glob
import json
import os

import pytest
import torch

from util import GOLD, surface

REF = os.environ.get("GG_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")


def _surfaces():
    with open(os.path.join(GOLD, "surfaces_full.json")) as f:
        return json.load(f)


def _sub(surf, prefix):
    return [[k[len(prefix):], s] for k, s in surf if k.startswith(prefix)]


def test_ccdm_params_eval_yml_builds_the_reference_surface():
    import yaml
    from jointimagegeneration_amd import ddpm_eval
    with open(os.path.join(REF, "ccdm", "params_eval.yml")) as f:
        params = yaml.safe_load(f)
    assert params["backbone"] == "unet_openai" and params["time_steps"] == 250 and params["evaluation_vote_strategy"] == "confidence"
    with torch.device("meta"):
        model = ddpm_eval.build_from_params(params, (128, 128, 128), 14)
    assert surface(model.unet) == _surfaces()["ccdm_full"]
    assert model.time_steps == 250 and model.step_T_sample == "confidence" and model.diffusion.num_classes == 14
    assert ddpm_eval.build_feature_cond_encoder(params) is None                       # feature_cond_encoder.type == 'none'


def test_every_latentdiffusion_yaml_instantiates_unedited():
    from jointimagegeneration_amd.config import instantiate_from_config, load_yaml
    from jointimagegeneration_amd.ldm import AutoencoderKL, IdentityEncoder, LatentDiffusion
    from jointimagegeneration_amd.sample_diffusion import strip_ckpt_paths
    files = sorted(glob.glob(os.path.join(REF, "latentdiffusion", "configs", "**", "*.yaml"), recursive=True))
    assert [os.path.basename(f) for f in files] == ["ruijin-pimage_and_mask_autoencoder_kl.yaml", "ruijin-ldm_from_controlnet.yaml",
                                                    "ruijin-ldm_from_controlnet_ae.yaml"]
    surf = _surfaces()
    nparams = lambda m: sum(p.numel() for p in m.parameters())
    built = {}
    for f in files:
        cfg = strip_ckpt_paths(load_yaml(f))                  # only the authors' /mnt/... ckpt_path strings are nulled (sample_diffusion.py)
        with torch.device("meta"):
            built[os.path.basename(f)] = instantiate_from_config(cfg["model"])
    # latent config (the C2 / C4 / C5 model): UNet + first stage surfaces equal the reference's
    m = built["ruijin-ldm_from_controlnet_ae.yaml"]
    assert isinstance(m, LatentDiffusion) and isinstance(m.first_stage_model, AutoencoderKL) and isinstance(m.cond_stage_model, AutoencoderKL)
    full = surface(m)
    assert _sub(full, "model.diffusion_model.") == surf["ldm_full"]
    assert _sub(full, "first_stage_model.") == surf["ae_full"]
    assert m.model.conditioning_key == "concat" and m.channels == 4 and m.image_size == 64 and m.num_timesteps == 1000
    assert abs(nparams(m.model.diffusion_model) - 267.5e6) < 0.1e6                    # 267.5 M
    assert any(k == "model_ema.diffusion_modeltime_embed0weight" for k, _ in full)    # LitEma name mangling (ema.py:15-22)
    # pixel-space config: no first stage, IdentityEncoder cond stage, UNet on 3 x 512^2
    p = built["ruijin-ldm_from_controlnet.yaml"]
    assert isinstance(p, LatentDiffusion) and p.no_first_stage and isinstance(p.cond_stage_model, IdentityEncoder)
    u = p.model.diffusion_model
    assert (u.in_channels, u.out_channels, u.model_channels) == (3, 1, 128) and abs(nparams(u) - 172.9e6) < 0.1e6   # 172.9 M
    # stand-alone autoencoder config
    a = built["ruijin-pimage_and_mask_autoencoder_kl.yaml"]
    assert isinstance(a, AutoencoderKL)
    assert [k for k, _ in surface(a)][:2] == ["encoder.conv_in.weight", "encoder.conv_in.bias"]
