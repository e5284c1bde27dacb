import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
torch.set_grad_enabled(False)
from oracle import samplers as S
from test_oracle_golden import build_small
from util import AE_SMALL, LDM_SMALL, T, gold, rel_err, rms_err, seeded
from jointimagegeneration_amd import ops
dev = torch.device("cuda:0")
for hint in (0, 1):
    ops.PATH_HINT = hint
    g = gold("networks_small")
    K, u, u2, u3, ae = build_small()
    u, u2, u3, ae = u.to(dev), u2.to(dev), u3.to(dev), ae.to(dev)
    lab = T(g["ccdm_labels"]).long()
    out = u(S.one_hot_bchw(lab, K).to(dev), torch.zeros(1, 1, 8, 8, 8, device=dev), None, T(g["ccdm_t"]).to(dev))["diffusion_out"]
    print(hint, "ccdm probs max abs", float((out.cpu() - T(g["ccdm_probs"])).abs().max()))
    e = u2(T(g["ldm_x"]).to(dev), T(g["ldm_t"]).to(dev)); print(hint, "ldm eps", rel_err(e, T(g["ldm_eps"])), rms_err(e, T(g["ldm_eps"])))
    e = u3(T(g["ldm_x"]).to(dev), T(g["ldm_t"]).to(dev), context=T(g["ldmst_ctx"]).to(dev)); print(hint, "ldmst eps", rel_err(e, T(g["ldmst_eps"])), rms_err(e, T(g["ldmst_eps"])))
    dec = ae.decode(T(g["ae_z"]).to(dev)); print(hint, "ae dec", rel_err(dec, T(g["ae_dec"])), rms_err(dec, T(g["ae_dec"])))
    mode = ae.encode(T(g["ae_img"]).to(dev)).mode(); print(hint, "ae mode", rel_err(mode, T(g["ae_mode"])), rms_err(mode, T(g["ae_mode"])))
ops.PATH_HINT = 0
from jointimagegeneration_amd.ldm import DDIMSampler, LatentDiffusion, PLMSSampler
g = gold("chains_small")
cfg_unet = dict(target="ldm.modules.diffusionmodules.openaimodel.UNetModel", params=dict(LDM_SMALL))
cfg_ae = dict(target="ldm.models.autoencoder.AutoencoderKL", params=dict(embed_dim=4, dims=2, ddconfig=dict(AE_SMALL), lossconfig=dict(target="torch.nn.Identity")))
cfg_cond = dict(target="ldm.models.autoencoder.AutoencoderKL", params=dict(embed_dim=4, dims=2, ddconfig=dict(AE_SMALL, in_channels=2, out_ch=2), lossconfig=dict(target="torch.nn.Identity")))
m = seeded(LatentDiffusion(first_stage_config=cfg_ae, cond_stage_config=cfg_cond, unet_config=cfg_unet, linear_start=0.0015, linear_end=0.0195, timesteps=1000, image_size=8, channels=4, dims=2, first_stage_key="image", cond_stage_key="mask", num_timesteps_cond=1), "ldm_pipe.").to(dev)
c = m.get_learned_conditioning(T(g["ldm_concat_cond"]).to(dev)); print("cond", rel_err(c, T(g["ldm_c"])), rms_err(c, T(g["ldm_c"])))
z, _ = DDIMSampler(m).sample(S=5, batch_size=2, shape=(4, 8, 8), conditioning=c, verbose=False, x_T=T(g["ldm_x_T"]).to(dev), dims=2, noise_tape=list(T(g["ldm_noises"])))
print("ddim z", rel_err(z, T(g["ldm_z"])), rms_err(z, T(g["ldm_z"])))
dec = m.decode_first_stage(z); print("dec", rel_err(dec, T(g["ldm_dec"])), rms_err(dec, T(g["ldm_dec"])))
zp, _ = PLMSSampler(m).sample(S=10, batch_size=2, shape=(4, 8, 8), conditioning=c, verbose=False, x_T=T(g["ldm_x_T"]).to(dev))
print("plms", rel_err(zp, T(g["ldm_plms_z"])), rms_err(zp, T(g["ldm_plms_z"])))
m20 = seeded(LatentDiffusion(first_stage_config=cfg_ae, cond_stage_config=cfg_cond, unet_config=cfg_unet, linear_start=0.0015, linear_end=0.0195, timesteps=20, image_size=8, channels=4, dims=2, first_stage_key="image", cond_stage_key="mask", num_timesteps_cond=1), "ldm_pipe.").to(dev)
zv = m20.p_sample_loop(c, (2, 4, 8, 8), x_T=T(g["ldm_x_T"]).to(dev), verbose=False, noise_tape=list(T(g["ldm_vanilla_noises"])))
print("vanilla", rel_err(zv, T(g["ldm_vanilla_z"])), rms_err(zv, T(g["ldm_vanilla_z"])))
