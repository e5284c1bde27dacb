"""Generates tests/golden/*.npz by importing the REFERENCE hot-path modules on CPU.

Run only in the build container (needs /root/reference):  python tests/golden/make_golden.py
The reference never travels: fixtures hold inputs/outputs (data) only; weights are
regenerated on both sides from jointimagegeneration_amd.synth (name, shape, seed).
While generating, every fixture is also cross-checked against the oracle restatement
(oracle/), so a drifted oracle fails here, loudly, before any fixture is written.
"""
from __future__ import annotations

import importlib
import json
import math
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = os.environ.get("GG_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))

from jointimagegeneration_amd.synth import randomize_parameters  # noqa: E402
from oracle import nets as O  # noqa: E402
from oracle import samplers as S  # noqa: E402

torch.set_grad_enabled(False)
SEED = 1024


# ----------------------------------------------------------------------------- reference import recipes
def import_ccdm():
    for name, path in (("ddpm", f"{REF}/ccdm/ddpm"), ("ddpm.models", f"{REF}/ccdm/ddpm/models")):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    dd = importlib.import_module("ddpm.models.diffusion_denoising")
    oh = importlib.import_module("ddpm.models.one_hot_categorical")
    un = importlib.import_module("ddpm.models.unet_openai")
    unet = importlib.import_module("ddpm.models.unet_openai.unet")
    nn_ = importlib.import_module("ddpm.models.unet_openai.nn")
    return dd, oh, un, unet, nn_


def import_ldm():
    sys.path.insert(0, f"{REF}/latentdiffusion")

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        sys.modules[name] = m
        return m

    def instantiate_from_config(config):
        if config in ("__is_first_stage__", "__is_unconditional__"):
            return None
        module, cls = config["target"].rsplit(".", 1)
        return getattr(importlib.import_module(module), cls)(**config.get("params", dict()))

    def exists(x):
        return x is not None

    def default(val, d):
        return val if val is not None else (d() if callable(d) else d)

    mod("models")
    mod("models.util", instantiate_from_config=instantiate_from_config, exists=exists, default=default,
        ismap=lambda x: False, isimage=lambda x: False, mean_flat=lambda t: t.mean(dim=list(range(1, t.ndim))),
        count_params=lambda m, verbose=False: sum(p.numel() for p in m.parameters()), log_txt_as_img=None)

    class LightningModule(torch.nn.Module):
        @property
        def device(self):
            return next(self.parameters()).device

    pl = mod("pytorch_lightning", LightningModule=LightningModule)
    mod("pytorch_lightning.utilities")
    mod("pytorch_lightning.utilities.distributed", rank_zero_only=lambda f: f)
    pl.utilities = sys.modules["pytorch_lightning.utilities"]
    mod("taming"); mod("taming.modules"); mod("taming.modules.vqvae")
    mod("taming.modules.vqvae.quantize", VectorQuantizer=object)
    mod("torchvision"); mod("torchvision.utils", make_grid=None)

    class ListConfig(list):
        pass
    mod("omegaconf"); mod("omegaconf.listconfig", ListConfig=ListConfig)

    om = importlib.import_module("ldm.modules.diffusionmodules.openaimodel")
    at = importlib.import_module("ldm.modules.attention")
    mo = importlib.import_module("ldm.modules.diffusionmodules.model")
    ae = importlib.import_module("ldm.models.autoencoder")
    dm = importlib.import_module("ldm.models.diffusion.ddpm")
    di = importlib.import_module("ldm.models.diffusion.ddim")
    ut = importlib.import_module("ldm.modules.diffusionmodules.util")
    di.DDIMSampler.register_buffer = lambda self, name, attr: setattr(self, name, attr)  # ddim.py:18-22 hard-codes cuda
    return om, at, mo, ae, dm, di, ut


def sd_of(m):
    return {k: v.detach().clone() for k, v in m.state_dict().items()}


def surface(m):
    return json.dumps([[k, list(v.shape)] for k, v in m.state_dict().items()])


def close(a, b, tol, what):
    err = (a - b).abs().max().item()
    ref = b.abs().max().item() + 1e-30
    assert err <= tol * max(1.0, ref), f"oracle drift on {what}: max|d|={err:.3e} (ref max {ref:.3e})"
    return err


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    sz = os.path.getsize(os.path.join(OUT, name + ".npz"))
    print(f"  wrote {name}.npz  ({sz/1024:.1f} KiB)")


def g(seed):
    return torch.Generator().manual_seed(seed)


# ----------------------------------------------------------------------------- fixtures
def fx_schedules(dd, dm_ut):
    _, _, _, _, dm, di, ut = dm_ut
    out = {}
    for T in (50, 250):
        b, a, c = dd.cosine_schedule(T)
        ob, oa, oc = S.ccdm_cosine_schedule(T)
        assert torch.equal(b, ob) and torch.equal(a, oa) and torch.equal(c, oc)
        out[f"cos{T}_betas"], out[f"cos{T}_alphas"], out[f"cos{T}_cumalphas"] = b, a, c
    b, a, c = dd.linear_schedule(50)
    ob, oa, oc = S.ccdm_linear_schedule(50)
    assert torch.equal(b, ob) and torch.equal(c, oc)
    out["lin50_betas"], out["lin50_cumalphas"] = b, c
    # known answers quoted in SURVEY 8a
    assert abs(float(out["cos250_betas"][0]) - 1.942687958944589e-4) < 1e-12
    assert abs(float(out["cos50_cumalphas"][1]) - 0.9980973601341248) < 1e-9
    betas = ut.make_beta_schedule("linear", 1000, linear_start=0.0015, linear_end=0.0195)
    assert np.array_equal(betas, S.ldm_linear_betas(1000, 0.0015, 0.0195))
    ac = np.cumprod(1.0 - betas, axis=0)
    ac32 = torch.tensor(ac, dtype=torch.float32)
    out["ldm_alphas_cumprod"] = ac32
    ts = ut.make_ddim_timesteps("uniform", 50, 1000, verbose=False)
    sig, al, alp = ut.make_ddim_sampling_parameters(ac32.cpu(), ts, 0.0, verbose=False)
    osch = S.ddim_schedule(ac32, 50, 0.0)
    assert np.array_equal(ts, osch["timesteps"])
    assert torch.equal(torch.as_tensor(al).float(), torch.as_tensor(osch["alphas"]).float())
    assert np.allclose(np.asarray(alp, dtype=np.float64), np.asarray(osch["alphas_prev"], dtype=np.float64), rtol=0, atol=0)
    out["ddim50_timesteps"] = ts
    out["ddim50_alphas"] = torch.as_tensor(al).float()
    out["ddim50_alphas_prev"] = torch.as_tensor(np.asarray(alp)).float()
    out["ddim50_sqrt_one_minus_alphas"] = torch.as_tensor(np.sqrt(1.0 - al)).float()
    assert abs(float(out["ddim50_alphas"][0]) - 0.9969944357872009) < 1e-9
    save("schedules", **out)


def fx_posterior(dd, oh):
    out = {}
    K, T = 14, 50
    dm_ = dd.DiffusionModel("cosine", T, K, dims=3)
    gen = g(7)
    shape = (2, K, 3, 4, 5)
    mism_total = 0
    for t in (1, 2, 25, 50):
        lab = torch.randint(0, K, (2, 3, 4, 5), generator=gen)
        xt = S.one_hot_bchw(lab, K)
        p0 = torch.softmax(2.0 * torch.randn(shape, generator=gen), dim=1)
        tt = torch.full((2,), t)
        ref = dm_.theta_post_prob(xt, p0, tt)
        a, abar = S.ccdm_step_scalars(dm_.alphas, dm_.cumalphas, t)
        mine = S.theta_post_prob(xt, p0, a, abar)
        close(mine, ref, 2e-6, f"theta_post_prob t={t}")
        # sampling: the reference draws under a seed; the tape is the same exponential_ call
        probs = torch.clamp(ref, min=1e-12)
        torch.manual_seed(100 + t)
        ref_onehot = oh.OneHotCategoricalBCHW(probs=probs).sample()
        ref_lab = ref_onehot.argmax(dim=1)
        E = torch.empty(2 * 3 * 4 * 5, K).exponential_(1, generator=g(100 + t))
        my_lab = S.race_sample_labels(torch.clamp(mine, min=1e-12), E)
        mism_total += int((my_lab != ref_lab).sum())
        out[f"t{t}_xt_labels"], out[f"t{t}_p0"], out[f"t{t}_probs"] = lab.int(), p0, ref
        out[f"t{t}_E"], out[f"t{t}_sample_labels"] = E, ref_lab.int()
        out[f"t{t}_a_abar"] = np.array([a, abar], dtype=np.float64)
    assert mism_total == 0, f"race-sampling restatement disagrees with the reference on {mism_total} voxels"
    # K=3 known answer
    dm3 = dd.DiffusionModel("cosine", 50, 3, dims=3)
    xt = S.one_hot_bchw(torch.ones(1, 1, 1, 1, dtype=torch.long), 3)
    p0 = torch.tensor([0.7, 0.2, 0.1]).reshape(1, 3, 1, 1, 1)
    r25 = dm3.theta_post_prob(xt, p0, torch.tensor([25])).flatten()
    assert torch.allclose(r25, torch.tensor([0.06401673, 0.91347396, 0.02250933]), atol=1e-7)
    out["k3_t25"] = r25
    save("ccdm_posterior", **out)


def fx_timestep_embedding(nn_ccdm, ldm_ut):
    t_f = torch.tensor([1.0, 17.0, 250.0])
    t_i = torch.tensor([1, 481, 981])
    a = nn_ccdm.timestep_embedding(t_f, 64)
    b = ldm_ut.timestep_embedding(t_i, 160)
    assert torch.equal(a, O.timestep_embedding(t_f, 64)) and torch.equal(b, O.timestep_embedding(t_i, 160))
    save("timestep_embedding", t_f=t_f, emb_f=a, t_i=t_i, emb_i=b)


def fx_modules(unet_ccdm, ldm):
    om, at, mo, ae, dm, di, ut = ldm
    out = {}
    gen = g(11)
    # --- CCDM 3D ResBlock (64 -> 96, 1x1 skip), emb 128
    rb = unet_ccdm.ResBlock(64, 128, 0.0, out_channels=96, dims=3).eval()
    randomize_parameters(rb, SEED, "rb3d.")
    x = torch.randn(1, 64, 4, 6, 8, generator=gen); emb = torch.randn(1, 128, generator=gen)
    y = rb(x, emb)
    sd = {"b." + k: v for k, v in sd_of(rb).items()}
    close(O.resblock(sd, "b.", x, emb), y, 1e-5, "ResBlock3d")
    out.update(rb3d_x=x, rb3d_emb=emb, rb3d_y=y); out["rb3d_surface"] = surface(rb)
    # --- CCDM 3D AttentionBlock (64 ch, head 32)
    ab = unet_ccdm.AttentionBlock(64, num_heads=1, num_head_channels=32).eval()
    randomize_parameters(ab, SEED, "ab3d.")
    x = torch.randn(2, 64, 2, 4, 4, generator=gen)
    y = ab(x)
    sd = {"b." + k: v for k, v in sd_of(ab).items()}
    close(O.attention_block(sd, "b.", x, 2), y, 1e-5, "AttentionBlock3d")
    out.update(ab3d_x=x, ab3d_y=y); out["ab3d_surface"] = surface(ab)
    # --- CCDM Up/Downsample 3D
    up = unet_ccdm.Upsample(32, True, dims=3).eval(); randomize_parameters(up, SEED, "up3d.")
    dn = unet_ccdm.Downsample(32, True, dims=3).eval(); randomize_parameters(dn, SEED, "dn3d.")
    x = torch.randn(1, 32, 4, 4, 6, generator=gen)
    yu, yd = up(x), dn(x)
    close(O.conv(O.upsample_nearest2(x), up.conv.weight, up.conv.bias, padding=1), yu, 1e-5, "Upsample3d")
    close(O.conv(x, dn.op.weight, dn.op.bias, stride=2, padding=1), yd, 1e-5, "Downsample3d")
    out.update(ud3d_x=x, up3d_y=yu, dn3d_y=yd)
    # --- LDM 2D ResBlock (identity skip) + AttentionBlock
    rb2 = om.ResBlock(64, 128, 0.0, out_channels=64, dims=2).eval(); randomize_parameters(rb2, SEED, "rb2d.")
    x = torch.randn(2, 64, 8, 8, generator=gen); emb = torch.randn(2, 128, generator=gen)
    y = rb2(x, emb)
    sd = {"b." + k: v for k, v in sd_of(rb2).items()}
    close(O.resblock(sd, "b.", x, emb), y, 1e-5, "ResBlock2d")
    out.update(rb2d_x=x, rb2d_emb=emb, rb2d_y=y)
    ab2 = om.AttentionBlock(96, num_heads=-1, num_head_channels=32).eval(); randomize_parameters(ab2, SEED, "ab2d.")
    x = torch.randn(2, 96, 8, 8, generator=gen)
    y = ab2(x)
    sd = {"b." + k: v for k, v in sd_of(ab2).items()}
    close(O.attention_block(sd, "b.", x, 3), y, 1e-5, "AttentionBlock2d")
    out.update(ab2d_x=x, ab2d_y=y)
    # --- SpatialTransformer with context
    st = at.SpatialTransformer(64, 2, 32, depth=1, context_dim=48).eval(); randomize_parameters(st, SEED, "st.")
    x = torch.randn(2, 64, 8, 8, generator=gen); ctx = torch.randn(2, 7, 48, generator=gen)
    y = st(x, ctx)
    sd = {"b." + k: v for k, v in sd_of(st).items()}
    close(O.spatial_transformer(sd, "b.", x, ctx, 2), y, 1e-5, "SpatialTransformer")
    out.update(st_x=x, st_ctx=ctx, st_y=y); out["st_surface"] = surface(st)
    # --- AE ResnetBlock / AttnBlock2d / Down / Up
    r = mo.ResnetBlock(in_channels=32, out_channels=64, dropout=0.0, temb_channels=0, dims=2).eval()
    randomize_parameters(r, SEED, "aer.")
    x = torch.randn(1, 32, 8, 8, generator=gen)
    y = r(x, None)
    sd = {"b." + k: v for k, v in sd_of(r).items()}
    close(O.ae_resnet_block(sd, "b.", x), y, 1e-5, "AE ResnetBlock")
    out.update(aer_x=x, aer_y=y)
    a2 = mo.AttnBlock2d(64).eval(); randomize_parameters(a2, SEED, "aea.")
    x = torch.randn(1, 64, 8, 8, generator=gen)
    y = a2(x)
    sd = {"b." + k: v for k, v in sd_of(a2).items()}
    close(O.ae_attn_block(sd, "b.", x), y, 1e-5, "AE AttnBlock2d")
    out.update(aea_x=x, aea_y=y)
    save("modules", **out)


CCDM_SMALL = dict(base_channels=32, channel_mult=[1, 2, 2], attention_resolutions=[2, 4], num_heads=1,
                  num_head_channels=32, softmax_output=True)
LDM_SMALL = dict(image_size=16, in_channels=8, out_channels=4, model_channels=32, attention_resolutions=[2, 4],
                 num_res_blocks=2, channel_mult=[1, 2, 2], num_head_channels=32, dims=2)
AE_SMALL = dict(double_z=True, z_channels=4, resolution=32, in_channels=1, out_ch=1, ch=32, ch_mult=[1, 2, 2],
                num_res_blocks=1, dropout=0.0, dims=2, attn_resolutions=[])


def fx_unets(un, ldm):
    om, at, mo, ae, dm, di, ut = ldm
    out = {}
    gen = g(13)
    K = 6
    u = un.create_unet_openai(image_size=16, in_channels=K + 1, out_channels=K, num_res_blocks=2,
                              cond_encoded_shape=None, dims=3, **CCDM_SMALL).eval()
    randomize_parameters(u, SEED, "ccdm_small.")
    lab = torch.randint(0, K, (1, 8, 8, 8), generator=gen)
    xt = S.one_hot_bchw(lab, K)
    cond = torch.zeros(1, 1, 8, 8, 8)
    t = torch.tensor([17.0])
    y = u(xt, cond, None, t)["diffusion_out"]
    mine = O.unet_forward(sd_of(u), torch.cat([xt, cond], 1), t, model_channels=32, head_channels=32, softmax_out=True)
    close(mine, y, 2e-5, "CCDM small UNet")
    out.update(ccdm_labels=lab.int(), ccdm_t=t, ccdm_probs=y); out["ccdm_surface"] = surface(u)
    # LDM small
    u2 = om.UNetModel(**LDM_SMALL).eval(); randomize_parameters(u2, SEED, "ldm_small.")
    x = torch.randn(2, 8, 16, 16, generator=gen); t = torch.tensor([981, 981])
    y = u2(x, t)
    mine = O.unet_forward(sd_of(u2), x, t, model_channels=32, head_channels=32)
    close(mine, y, 2e-5, "LDM small UNet")
    out.update(ldm_x=x, ldm_t=t, ldm_eps=y); out["ldm_surface"] = surface(u2)
    # LDM small with SpatialTransformer
    cfg = dict(LDM_SMALL, use_spatial_transformer=True, transformer_depth=1, context_dim=48)
    u3 = om.UNetModel(**cfg).eval(); randomize_parameters(u3, SEED, "ldm_small_st.")
    ctx = torch.randn(2, 5, 48, generator=gen)
    y = u3(x, t, context=ctx)
    mine = O.unet_forward(sd_of(u3), x, t, model_channels=32, head_channels=32, context=ctx)
    close(mine, y, 2e-5, "LDM small UNet + SpatialTransformer")
    out.update(ldmst_ctx=ctx, ldmst_eps=y); out["ldmst_surface"] = surface(u3)
    # AE small
    a = ae.AutoencoderKL(ddconfig=AE_SMALL, lossconfig=dict(target="torch.nn.Identity"), embed_dim=4, dims=2).eval()
    randomize_parameters(a, SEED, "ae_small.")
    z = torch.randn(1, 4, 8, 8, generator=gen)
    img = torch.randn(1, 1, 32, 32, generator=gen)
    dec = a.decode(z)
    mode = a.encode(img).mode()
    sd = sd_of(a)
    close(O.ae_decode(sd, z), dec, 2e-5, "AE decode")
    close(O.ae_encode_mode(sd, img), mode, 2e-5, "AE encode.mode")
    out.update(ae_z=z, ae_img=img, ae_dec=dec, ae_mode=mode); out["ae_surface"] = surface(a)
    save("networks_small", **out)


def fx_chains(dd, oh, un, ldm):
    om, at, mo, ae, dm, di, ut = ldm
    out = {}
    # ---- CCDM: 5-step chain on the small UNet, unmodified reference loop under torch.manual_seed
    K, T = 6, 5
    u = un.create_unet_openai(image_size=16, in_channels=K + 1, out_channels=K, num_res_blocks=2,
                              cond_encoded_shape=None, dims=3, **CCDM_SMALL).eval()
    randomize_parameters(u, SEED, "ccdm_small.")
    model = dd.DenoisingModel(dd.DiffusionModel("cosine", T, K, dims=3), u, "none", "confidence", dims=3).eval()
    torch.manual_seed(SEED)
    x_T = oh.OneHotCategoricalBCHW(logits=torch.zeros(1, K, 8, 8, 8)).sample()   # evaluator.py:135-136
    cond = torch.zeros(1, 1, 8, 8, 8)
    ref_probs = model(x_T, cond)["diffusion_out"]
    ref_lab = ref_probs.argmax(dim=1)
    gen = g(SEED)
    M = 8 * 8 * 8
    E0 = torch.empty(M, K).exponential_(1, generator=gen)
    tapes = [torch.empty(M, K).exponential_(1, generator=gen) for _ in range(T - 1)]
    xT_lab = S.race_sample_labels(torch.full((1, K, 8, 8, 8), 1.0 / K), E0)
    assert torch.equal(xT_lab, x_T.argmax(dim=1)), "x_T tape does not reproduce the reference draw"
    sd = sd_of(u)

    def unet_probs(xt, t):
        return O.unet_forward(sd, torch.cat([xt, cond], 1), torch.tensor([t]), model_channels=32, head_channels=32,
                              softmax_out=True)
    trace = []
    my_lab, my_probs = S.ccdm_chain(unet_probs, xT_lab, K, "cosine", T, tapes, "confidence", trace=trace)
    mism = int((my_lab != ref_lab).sum())
    print(f"  CCDM small chain: label mismatches vs reference = {mism}/{M}")
    assert mism == 0
    close(my_probs, ref_probs, 1e-4, "CCDM chain final probs")
    out.update(ccdm_E0=E0, ccdm_tapes=torch.stack(tapes), ccdm_final_labels=ref_lab.int(), ccdm_final_probs=ref_probs)
    out["ccdm_step_labels"] = torch.stack([tr["labels"] for tr in trace]).int()
    # ---- LDM: LatentDiffusion + DDIMSampler 5 steps + cond-encode + decode, small config
    cfg_unet = dict(target="ldm.modules.diffusionmodules.openaimodel.UNetModel", params=dict(LDM_SMALL))
    cfg_ae = dict(target="ldm.models.autoencoder.AutoencoderKL",
                  params=dict(embed_dim=4, dims=2, ddconfig=dict(AE_SMALL), lossconfig=dict(target="torch.nn.Identity")))
    cfg_cond = dict(target="ldm.models.autoencoder.AutoencoderKL",
                    params=dict(embed_dim=4, dims=2, ddconfig=dict(AE_SMALL, in_channels=2, out_ch=2),
                                lossconfig=dict(target="torch.nn.Identity")))
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        m = dm.LatentDiffusion(first_stage_config=cfg_ae, cond_stage_config=cfg_cond, unet_config=cfg_unet,
                               linear_start=0.0015, linear_end=0.0195, timesteps=1000, image_size=8, channels=4, dims=2,
                               first_stage_key="image", cond_stage_key="mask", num_timesteps_cond=1).eval()
    randomize_parameters(m, SEED, "ldm_pipe.")
    gen = g(2048)
    concat_cond = torch.rand(2, 2, 32, 32, generator=gen)
    x_T = torch.randn(2, 4, 8, 8, generator=gen)
    noises = [torch.randn(2, 4, 8, 8, generator=gen) for _ in range(5)]
    c = m.get_learned_conditioning(concat_cond)
    sampler = di.DDIMSampler(m)
    # feed the tape through the global generator in the order ddim.py:124,201 consumes it
    import unittest.mock as mock
    tape = iter(noises)
    real_randn = torch.randn

    def fake_randn(*a, **k):
        return next(tape)
    with mock.patch.object(ut.torch, "randn", fake_randn):
        z, _ = sampler.sample(S=5, batch_size=2, shape=(4, 8, 8), conditioning=c, verbose=False, x_T=x_T, dims=2)
    dec = m.decode_first_stage(z)
    sd_all = sd_of(m)
    sd_unet = O.sub_state_dict(sd_all, "model.diffusion_model.")
    sd_fs = O.sub_state_dict(sd_all, "first_stage_model.")
    sd_cs = O.sub_state_dict(sd_all, "cond_stage_model.")
    my_c = O.ae_encode_mode(sd_cs, concat_cond)
    close(my_c, c, 2e-5, "cond encode")

    def eps(x, t):
        return O.unet_forward(sd_unet, torch.cat([x, my_c], 1), t, model_channels=32, head_channels=32)
    my_z, _ = S.ddim_sample(eps, x_T, noises, m.alphas_cumprod, 5)
    close(my_z, z, 1e-4, "DDIM 5-step latent")
    my_dec = O.ae_decode(sd_fs, my_z)
    close(my_dec, dec, 2e-4, "decode_first_stage")
    # ---- PLMS, 10 steps (exercises the Euler start and Adams-Bashforth orders 2..4), same model / conditioning
    pl = importlib.import_module("ldm.models.diffusion.plms")
    pl.PLMSSampler.register_buffer = lambda self, name, attr: setattr(self, name, attr)
    ps = pl.PLMSSampler(m)
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        with mock.patch.object(ut.torch, "randn", lambda *a, **k: torch.zeros(2, 4, 8, 8)):
            zp, _ = ps.sample(S=10, batch_size=2, shape=(4, 8, 8), conditioning=c, verbose=False, x_T=x_T)
    my_zp, _ = S.plms_sample(eps, x_T, m.alphas_cumprod, 10)
    close(my_zp, zp, 2e-4, "PLMS 10-step latent")
    out.update(ldm_plms_z=zp)
    # ---- vanilla ancestral sampling (p_sample_loop), a 20-step schedule with the same networks and conditioning
    with contextlib.redirect_stdout(io.StringIO()):
        m20 = dm.LatentDiffusion(first_stage_config=cfg_ae, cond_stage_config=cfg_cond, unet_config=cfg_unet,
                                 linear_start=0.0015, linear_end=0.0195, timesteps=20, image_size=8, channels=4, dims=2,
                                 first_stage_key="image", cond_stage_key="mask", num_timesteps_cond=1).eval()
    randomize_parameters(m20, SEED, "ldm_pipe.")
    gen20 = g(4096)
    nz20 = [torch.randn(2, 4, 8, 8, generator=gen20) for _ in range(20)]
    tape20 = iter(nz20)
    with mock.patch.object(ut.torch, "randn", lambda *a, **k: next(tape20)):
        zv = m20.p_sample_loop(c, (2, 4, 8, 8), x_T=x_T, verbose=False)
    my_zv = S.ddpm_ancestral_sample(eps, x_T, nz20, S.ldm_linear_betas(20, 0.0015, 0.0195))
    close(my_zv, zv, 2e-4, "vanilla 20-step latent")
    out.update(ldm_vanilla_z=zv, ldm_vanilla_noises=torch.stack(nz20))
    out.update(ldm_concat_cond=concat_cond, ldm_x_T=x_T, ldm_noises=torch.stack(noises), ldm_c=c, ldm_z=z, ldm_dec=dec)
    out["ldm_pipe_surface"] = surface(m)
    save("chains_small", **out)


def fx_full_surfaces(un, ldm):
    """state_dict surface (names+shapes) of the full-size modules, built on the meta device."""
    om, at, mo, ae, dm, di, ut = ldm
    out = {}
    with torch.device("meta"):
        u = un.create_unet_openai(image_size=128, in_channels=15, out_channels=14, num_res_blocks=2,
                                  cond_encoded_shape=None, dims=3, base_channels=64, channel_mult=[1, 2, 2, 4, 5],
                                  attention_resolutions=[32, 16, 8], num_heads=1, num_head_channels=32, softmax_output=True)
        out["ccdm_full"] = surface(u)
        u2 = om.UNetModel(dims=2, image_size=512, in_channels=8, out_channels=4, model_channels=160,
                          attention_resolutions=[8, 4, 2], num_res_blocks=2, channel_mult=[1, 2, 4, 4, 5],
                          num_head_channels=32)
        out["ldm_full"] = surface(u2)
        import io, contextlib
        with contextlib.redirect_stdout(io.StringIO()):
            a = ae.AutoencoderKL(ddconfig=dict(double_z=True, z_channels=4, resolution=512, in_channels=1, out_ch=1,
                                               ch=128, ch_mult=[1, 2, 4, 4], num_res_blocks=2, dropout=0.0, dims=2,
                                               attn_resolutions=[16, 8]),
                                 lossconfig=dict(target="torch.nn.Identity"), embed_dim=4, dims=2)
        out["ae_full"] = surface(a)
    with open(os.path.join(OUT, "surfaces_full.json"), "w") as f:
        json.dump({k: json.loads(v) for k, v in out.items()}, f)
    print("  wrote surfaces_full.json")


def fx_e2e(dd, oh, un, ldm, which):
    """BASELINE configs C1 (CCDM 32^3, K=14, T=50, N=1) and C2 (LDM N=4, 4x32x32, 50 steps), full-size networks."""
    om, at, mo, ae, dm, di, ut = ldm
    if "c1" in which:
        K, T, R = 14, 50, 32
        u = un.create_unet_openai(image_size=R, in_channels=K + 1, out_channels=K, num_res_blocks=2,
                                  cond_encoded_shape=None, dims=3, base_channels=64, channel_mult=[1, 2, 2, 4, 5],
                                  attention_resolutions=[32, 16, 8], num_heads=1, num_head_channels=32,
                                  softmax_output=True).eval()
        randomize_parameters(u, SEED, "ccdm.")
        model = dd.DenoisingModel(dd.DiffusionModel("cosine", T, K, dims=3), u, "none", "confidence", dims=3).eval()
        torch.manual_seed(SEED)
        x_T = oh.OneHotCategoricalBCHW(logits=torch.zeros(1, K, R, R, R)).sample()
        cond = torch.zeros(1, 1, R, R, R)
        # the reference loop hands x_t to the UNet once per step (diffusion_denoising.py:206): record it for teacher forcing
        states, orig_forward = [], u.forward

        def recording_forward(x, *a, **k):
            states.append(x.argmax(dim=1).to(torch.uint8).clone())
            return orig_forward(x, *a, **k)
        u.forward = recording_forward
        ref_lab = model(x_T, cond)["diffusion_out"].argmax(dim=1)
        u.forward = orig_forward
        assert len(states) == T and torch.equal(states[0].long(), x_T.argmax(dim=1))
        states.append(ref_lab.to(torch.uint8))                  # states[i] = x_{T-i}; states[T] = final labels
        gen = g(SEED); M = R ** 3
        E0 = torch.empty(M, K).exponential_(1, generator=gen)
        tapes = [torch.empty(M, K).exponential_(1, generator=gen) for _ in range(T - 1)]
        xT_lab = S.race_sample_labels(torch.full((1, K, R, R, R), 1.0 / K), E0)
        assert torch.equal(xT_lab, x_T.argmax(dim=1))
        sd = sd_of(u)

        def unet_probs(xt, t):
            return O.unet_forward(sd, torch.cat([xt, cond], 1), torch.tensor([t]), model_channels=64, head_channels=32,
                                  softmax_out=True)
        my_lab, _ = S.ccdm_chain(unet_probs, xT_lab, K, "cosine", T, tapes, "confidence")
        mism = int((my_lab != ref_lab).sum())
        print(f"  C1 chain: oracle-vs-reference label mismatches = {mism}/{M}")
        # teacher-forced single steps (first, middle, last sampled, final argmax): reference x_t in -> reference x_{t-1} out;
        # the oracle must reproduce each of them from the same input and the same tape before the fixture is written
        step_t = [T, T - 1, T // 2 + 1, 2, 1]
        _, al, ca = S.ccdm_schedule("cosine", T)
        tf_mism = []
        for t in step_t:
            i = T - t
            xin = states[i].long()
            xt = S.one_hot_bchw(xin, K)
            a, abar = S.ccdm_step_scalars(al, ca, t)
            pr = torch.clamp(S.theta_post_prob(xt, unet_probs(xt, float(t)), a, abar), min=1e-12)
            nxt = S.race_sample_labels(pr, tapes[i]) if t > 1 else pr.argmax(dim=1)
            tf_mism.append(int((nxt != states[i + 1].long()).sum()))
        print(f"  C1 teacher-forced oracle-vs-reference mismatches per step {step_t}: {tf_mism}")
        assert max(tf_mism) <= 2, tf_mism
        save("e2e_c1", labels=ref_lab.to(torch.uint8), oracle_mismatches=np.array(mism),
             hist=torch.bincount(ref_lab.flatten(), minlength=K), step_t=np.array(step_t),
             step_in=torch.stack([states[T - t][0] for t in step_t]), step_out=torch.stack([states[T - t + 1][0] for t in step_t]),
             step_oracle_mismatches=np.array(tf_mism))
    if "c2" in which:
        u2 = om.UNetModel(dims=2, image_size=512, in_channels=8, out_channels=4, model_channels=160,
                          attention_resolutions=[8, 4, 2], num_res_blocks=2, channel_mult=[1, 2, 4, 4, 5],
                          num_head_channels=32).eval()
        randomize_parameters(u2, SEED, "ldm.")
        gen = g(2048)
        c = torch.randn(4, 4, 32, 32, generator=gen)
        x_T = torch.randn(4, 4, 32, 32, generator=gen)
        ac32 = torch.tensor(np.cumprod(1.0 - ut.make_beta_schedule("linear", 1000, 0.0015, 0.0195)), dtype=torch.float32)
        sd = sd_of(u2)

        class Shim:  # minimal model surface DDIMSampler reads (ddim.py:15,27-33,121,173)
            num_timesteps = 1000
            betas = torch.zeros(1000); alphas_cumprod = ac32
            alphas_cumprod_prev = torch.cat([torch.ones(1), ac32[:-1]]); device = torch.device("cpu")
            def apply_model(self, x, t, cc):
                return u2(torch.cat([x, cc], 1), t)
        sampler = di.DDIMSampler(Shim())
        import unittest.mock as mock
        with mock.patch.object(ut.torch, "randn", lambda *a, **k: torch.zeros(4, 4, 32, 32)):  # eta=0: noise unused
            z, _ = sampler.sample(S=50, batch_size=4, shape=(4, 32, 32), conditioning=c, verbose=False, x_T=x_T, dims=2)

        def eps(x, t):
            return O.unet_forward(sd, torch.cat([x, c], 1), t, model_channels=160, head_channels=32)
        my_z, _ = S.ddim_sample(eps, x_T, [torch.zeros_like(x_T)] * 50, ac32, 50)
        err = close(my_z, z, 5e-4, "C2 latent")
        print(f"  C2: oracle-vs-reference max|d| = {err:.2e}")
        save("e2e_c2", z=z.half(), c_seed=np.array(2048))


def fx_autoreg(ldm):
    """B8: the autoregressive slice loop.  The reference's `sample_cond` itself cannot run here (hard-wired `.cuda()`, PNG dumps
    through PIL / torchvision into ./samples/layers, latentdiffusion/sample_diffusion.py:199-223), so its LOOP BODY (:206-222) is
    executed statement by statement on the imported reference LatentDiffusion / DDIMSampler objects: concat_cond -> 
    get_learned_conditioning -> sampler.sample -> decode_first_stage -> min-max normalise -> feed back; 11 slices, 5 DDIM steps,
    x_T / per-step noises from a tape in the order ddim.py:124,201 draws them.  oracle.samplers.autoregressive_slices must
    reproduce every slice before the fixture is written."""
    om, at, mo, ae, dm, di, ut = ldm
    import io, contextlib
    import unittest.mock as mock
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import synth_labels
    cfg_unet = dict(target="ldm.modules.diffusionmodules.openaimodel.UNetModel", params=dict(LDM_SMALL))
    cfg_ae = lambda cin: dict(target="ldm.models.autoencoder.AutoencoderKL",
                              params=dict(embed_dim=4, dims=2, ddconfig=dict(AE_SMALL, in_channels=cin, out_ch=cin),
                                          lossconfig=dict(target="torch.nn.Identity")))
    with contextlib.redirect_stdout(io.StringIO()):
        m = dm.LatentDiffusion(first_stage_config=cfg_ae(1), cond_stage_config=cfg_ae(2), unet_config=cfg_unet,
                               linear_start=0.0015, linear_end=0.0195, timesteps=1000, image_size=8, channels=4, dims=2,
                               first_stage_key="image", cond_stage_key="mask", num_timesteps_cond=1).eval()
    randomize_parameters(m, SEED, "ldm_pipe.")
    D, hw, S_steps, n_samples = 11, 32, 5, 1
    lab = synth_labels((D, hw, hw), 12, seed=21)
    lab[0] = 0                                                      # start_layer = 1: the loop begins at m = 0
    assert lab[1:].reshape(D - 1, -1).any(1).all()
    wholemask = (torch.from_numpy(lab).float() / 255.0)[None, None]                       # [1, 1, D, H, W]
    gen = g(8192)
    shape = (m.channels, m.image_size, m.image_size)
    tape = [torch.randn((n_samples,) + shape, generator=gen) for _ in range(D * (S_steps + 1))]        # slices m = start-1 .. end = 0 .. D-1
    # ---- reference loop body (sample_diffusion.py:201-222)
    sampler = di.DDIMSampler(m)
    start_layer, end_layer = torch.where(wholemask.sum((0, 1, 3, 4)))[0][[0, -1]]
    samples = torch.zeros((n_samples,) + wholemask.shape[1:], dtype=torch.float32)
    gen_mask = wholemask.repeat(n_samples, 1, 1, 1, 1)
    it = iter(tape)
    with mock.patch.object(ut.torch, "randn", lambda *a, **k: next(it)):
        for m_ in range(start_layer.item() - 1, end_layer.item() + 1):
            concat_cond = torch.cat([samples[:, :, max(0, m_ - 1)], gen_mask[:, :, m_]], axis=1)
            c = m.get_learned_conditioning(concat_cond)
            s, _ = sampler.sample(S=S_steps, dims=len(shape) - 1, conditioning=c, batch_size=n_samples, shape=shape, verbose=False)
            ds = m.decode_first_stage(s)
            samples[:, :, m_] = (ds - ds.min()) / (ds.max() - ds.min())
    assert next(it, None) is None, "the reference consumed fewer draws than the tape holds"
    # ---- oracle
    sd_all = sd_of(m)
    sd_unet, sd_fs, sd_cs = (O.sub_state_dict(sd_all, p) for p in ("model.diffusion_model.", "first_stage_model.", "cond_stage_model."))
    it2 = iter(tape)
    mine = S.autoregressive_slices(lambda cc: O.ae_encode_mode(sd_cs, cc),
                                   lambda c: (lambda x, t: O.unet_forward(sd_unet, torch.cat([x, c], 1), t, model_channels=32, head_channels=32)),
                                   lambda z: O.ae_decode(sd_fs, z), wholemask, n_samples, shape, lambda shp: next(it2), m.alphas_cumprod, S_steps)
    per_slice = (mine - samples).abs().flatten(3).max(-1).values[0, 0]
    print("  autoregressive slices: oracle-vs-reference max|d| per slice:", " ".join(f"{float(v):.1e}" for v in per_slice))
    close(mine, samples, 2e-4, "autoregressive slice loop")
    xT = torch.stack(tape[::S_steps + 1])                                                 # the x_T draws (eta = 0: step noises unused)
    save("autoreg_small", labels=torch.from_numpy(lab).to(torch.uint8), x_T=xT, samples=samples, ddim_steps=np.array(S_steps))


def fx_ddim_options(ldm):
    """Sampler options of the hot-path DDIMSampler that no shipped config uses (VERDICT r03 missing #3): the "quad" discretisation
    (util.py:50-52) and classifier-free guidance (ddim.py:175-180), both produced by the REFERENCE sampler on the small LatentDiffusion
    of fx_chains (same weights, same conditioning inputs), cross-checked against the oracle."""
    om, at, mo, ae, dm, di, ut = ldm
    import contextlib
    import io
    import unittest.mock as mock
    out = {}
    for S_, T_ in ((5, 1000), (10, 1000), (50, 1000), (7, 100)):
        ts = ut.make_ddim_timesteps("quad", S_, T_, verbose=False)
        assert np.array_equal(ts, S.ddim_timesteps("quad", S_, T_))
        out[f"quad_ts_{S_}_{T_}"] = ts
    cfg_unet = dict(target="ldm.modules.diffusionmodules.openaimodel.UNetModel", params=dict(LDM_SMALL))
    cfg_ae = dict(target="ldm.models.autoencoder.AutoencoderKL",
                  params=dict(embed_dim=4, dims=2, ddconfig=dict(AE_SMALL), lossconfig=dict(target="torch.nn.Identity")))
    cfg_cond = dict(target="ldm.models.autoencoder.AutoencoderKL",
                    params=dict(embed_dim=4, dims=2, ddconfig=dict(AE_SMALL, in_channels=2, out_ch=2), lossconfig=dict(target="torch.nn.Identity")))
    with contextlib.redirect_stdout(io.StringIO()):
        m = dm.LatentDiffusion(first_stage_config=cfg_ae, cond_stage_config=cfg_cond, unet_config=cfg_unet, linear_start=0.0015, linear_end=0.0195,
                               timesteps=1000, image_size=8, channels=4, dims=2, first_stage_key="image", cond_stage_key="mask",
                               num_timesteps_cond=1).eval()
    randomize_parameters(m, SEED, "ldm_pipe.")
    gen = g(2048)
    concat_cond = torch.rand(2, 2, 32, 32, generator=gen)
    x_T = torch.randn(2, 4, 8, 8, generator=gen)
    c = m.get_learned_conditioning(concat_cond)
    uc = m.get_learned_conditioning(torch.zeros_like(concat_cond))          # "no mask, no previous slice" as the unconditional input
    sd_all = sd_of(m)
    sd_unet = O.sub_state_dict(sd_all, "model.diffusion_model.")
    eps_c = lambda x, t: O.unet_forward(sd_unet, torch.cat([x, c], 1), t, model_channels=32, head_channels=32)
    eps_u = lambda x, t: O.unet_forward(sd_unet, torch.cat([x, uc], 1), t, model_channels=32, head_channels=32)
    zeros = [torch.zeros(2, 4, 8, 8)] * 8
    sampler = di.DDIMSampler(m)
    with mock.patch.object(ut.torch, "randn", lambda *a, **k: torch.zeros(2, 4, 8, 8)):
        # (i) guidance scale 3 on the uniform 5-step schedule, through the reference's sample()
        z_cfg, _ = sampler.sample(S=5, batch_size=2, shape=(4, 8, 8), conditioning=c, verbose=False, x_T=x_T, dims=2,
                                  unconditional_guidance_scale=3.0, unconditional_conditioning=uc)
        # (ii) the quad schedule: reachable in the reference through make_schedule + ddim_sampling only
        sampler.make_schedule(ddim_num_steps=6, ddim_discretize="quad", ddim_eta=0.0, verbose=False)
        z_quad, _ = sampler.ddim_sampling(c, (2, 4, 8, 8), dims=2, x_T=x_T)
    my_cfg, _ = S.ddim_sample(eps_c, x_T, zeros, m.alphas_cumprod, 5, eps_uncond=eps_u, guidance_scale=3.0)
    close(my_cfg, z_cfg, 2e-4, "DDIM 5 steps, guidance scale 3")
    my_quad, _ = S.ddim_sample(eps_c, x_T, zeros, m.alphas_cumprod, 6, discretize="quad")
    close(my_quad, z_quad, 2e-4, "DDIM 6 steps, quad schedule")
    out.update(concat_cond=concat_cond, x_T=x_T, c=c, uc=uc, z_cfg=z_cfg, z_quad=z_quad, quad_ts_used=sampler.ddim_timesteps)
    save("ddim_options", **out)


def fx_glue():
    """Stage glue: the recipe of latentdiffusion/sample_diffusion.py:199-200,
    rot90(scipy.ndimage.zoom(mask, target / shape, order=0), dims=(1, 2), k=3) / 255, run with scipy itself."""
    from scipy.ndimage import zoom
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import synth_labels
    out = {}
    pairs = [(128, 512), (128, 256), (64, 96), (64, 256), (10, 23), (12, 32), (14, 32), (7, 512), (100, 333), (5, 5), (1, 4)]
    for n_in, n_out in pairs:
        z = zoom(np.arange(1, n_in + 1, dtype=np.int64), n_out / n_in, order=0)
        assert z.shape == (n_out,) and z.min() >= 1, (n_in, n_out, z.shape)
        idx = z - 1
        assert np.array_equal(idx, S.zoom0_index(n_in, n_out)), f"oracle drift on zoom index {n_in}->{n_out}"
        out[f"idx_{n_in}_{n_out}"] = idx.astype(np.int32)
    out["pairs"] = np.array(pairs)

    def recipe(lab, target):
        zz = zoom(lab, np.array(target) / np.array(lab.shape), order=0)
        assert zz.shape == tuple(target)
        return torch.rot90(torch.from_numpy(zz), dims=(1, 2), k=3)
    # non-integer ratios on every axis, whole result kept
    lab = synth_labels((10, 12, 14), 12, seed=3)
    rot = recipe(lab, (23, 32, 32))
    assert torch.equal(rot.float() / 255.0, S.mask_to_cond_volume(torch.from_numpy(lab), (23, 32, 32)))
    out["small_rot_labels"] = rot.to(torch.uint8)
    # BASELINE size 128^3 -> 256 x 512 x 512: position-sensitive per-slice checksums (the GPU test also recomputes the
    # recipe with scipy on the box and compares every voxel)
    lab = synth_labels((128, 128, 128), 12, seed=7)
    rot = recipe(lab, (256, 512, 512)).long()
    assert torch.equal(rot.float() / 255.0, S.mask_to_cond_volume(torch.from_numpy(lab), (256, 512, 512)))
    i = torch.arange(512)[:, None]; j = torch.arange(512)[None, :]
    wgt = ((i * 7 + j * 13) % 31 + 1).long()
    out["full_slice_sum"] = rot.sum((1, 2))
    out["full_slice_wsum"] = (rot * wgt[None]).sum((1, 2))
    save("glue", **out)


def fx_text_encoder():
    """PreloadedBERTEncoder (ccdm/ddpm/models/encoder.py:103-123) at its shipped size (768, 8 heads x 64, depth 4)
    on cached-feature-shaped random inputs [2, 768, 40], eval mode; reference output + oracle cross-check."""
    enc = importlib.import_module("ddpm.models.encoder")
    m = enc.PreloadedBERTEncoder(embed_dim=768, n_heads=8, depth=4, d_head=64, dropout=0.1).eval()
    randomize_parameters(m, SEED, "bertenc.")
    feats = torch.randn(2, 768, 40, generator=g(31))
    ref = m(feats)
    mine = O.preloaded_bert_encoder(sd_of(m), feats, 8)
    err = close(mine, ref, 2e-5, "PreloadedBERTEncoder")
    print(f"  text encoder: oracle-vs-reference max|d| = {err:.2e}")
    save("text_encoder", feats=feats, out=ref, surface=np.array(surface(m)))


if __name__ == "__main__":
    which = set(sys.argv[1:]) or {"small"}
    dd, oh, un, unet_ccdm, nn_ccdm = import_ccdm()
    ldm = import_ldm()
    if "small" in which or "all" in which:
        print("schedules"); fx_schedules(dd, ldm)
        print("posterior"); fx_posterior(dd, oh)
        print("timestep_embedding"); fx_timestep_embedding(nn_ccdm, ldm[6])
        print("modules"); fx_modules(unet_ccdm, ldm)
        print("unets"); fx_unets(un, ldm)
        print("chains"); fx_chains(dd, oh, un, ldm)
        print("surfaces"); fx_full_surfaces(un, ldm)
    if "text" in which or "small" in which or "all" in which:
        print("text encoder"); fx_text_encoder()
    if "glue" in which or "small" in which or "all" in which:
        print("glue"); fx_glue()
    if "autoreg" in which or "small" in which or "all" in which:
        print("autoreg"); fx_autoreg(ldm)
    if "ddimopt" in which or "small" in which or "all" in which:
        print("ddim options"); fx_ddim_options(ldm)
    if "c1" in which or "all" in which:
        fx_e2e(dd, oh, un, ldm, {"c1"})
    if "c2" in which or "all" in which:
        fx_e2e(dd, oh, un, ldm, {"c2"})
