"""Per-block error budget of the CCDM UNet forward (bf16 production engine vs the fp32 oracle) at C1 size (95.4 M params, 32^3):
for every TimestepEmbedSequential block, (a) the error the block ADDS when it is fed the oracle's own input (teacher-forced:
oracle activation -> bf16 CL -> engine block, compared with the oracle's output of that block) and (b) the accumulated error of
the free-running engine at the same point.  Errors are relative to the rms of the oracle activation.  VERDICT r02 item 4a.
    python tools/ccdm_error_budget.py            (GPU box; ~1 minute)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from jointimagegeneration_amd import ops  # noqa: E402
from jointimagegeneration_amd.ops import CL  # noqa: E402
from jointimagegeneration_amd.synth import randomize_parameters  # noqa: E402
from jointimagegeneration_amd.unet import create_unet_openai  # noqa: E402
from oracle import nets as O  # noqa: E402
from oracle import samplers as S  # noqa: E402
from util import CCDM_FULL, synth_labels  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
K, R = 14, int(os.environ.get("GG_BUDGET_R", "32"))
u = create_unet_openai(image_size=128, in_channels=K + 1, out_channels=K, num_res_blocks=2, cond_encoded_shape=None, dims=3, **CCDM_FULL).eval()
randomize_parameters(u, 1024, "ccdm.")
sd = {k: v.detach().float().clone() for k, v in u.state_dict().items()}
u = u.to(dev)
lab = torch.from_numpy(synth_labels((R, R, R), K, seed=11))[None]
x = torch.cat([S.one_hot_bchw(lab, K), torch.zeros(1, 1, R, R, R)], 1)
t = torch.tensor([25.0])
torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))

emb = O.timestep_embedding(t, 64)
emb = torch.nn.functional.linear(emb, sd["time_embed.0.weight"], sd["time_embed.0.bias"])
emb = torch.nn.functional.linear(O.silu(emb), sd["time_embed.2.weight"], sd["time_embed.2.bias"])
bias_row = u.time_bias_rows(t.to(dev))
lay, _ = u.time_bias_layout(1)
tb = lambda rb: bias_row[lay[id(rb)][0]:lay[id(rb)][0] + lay[id(rb)][1]]


def cl(a):
    return ops.to_cl(a.to(dev))


def back(c):
    return ops.from_cl(c, 3).cpu()


def err(got, ref):
    d = got - ref
    rms = float(ref.pow(2).mean().sqrt())
    return float(d.pow(2).mean().sqrt()) / rms, float(d.abs().max()) / rms


rows = []
ops.stats_begin(dev)
hs_o, hs_e = [], []
ho, he = x, cl(x)
names = [f"input_blocks.{i}" for i in range(len(u.input_blocks))] + ["middle_block"] + [f"output_blocks.{i}" for i in range(len(u.output_blocks))]
mods = list(u.input_blocks) + [u.middle_block] + list(u.output_blocks)
for name, mod in zip(names, mods):
    is_out = name.startswith("output")
    skip_o = hs_o.pop() if is_out else None
    skip_e = hs_e.pop() if is_out else None
    hin_o = torch.cat([ho, skip_o], 1) if is_out else ho
    ref = O._run_sequential(sd, name + ".", hin_o, emb, None, 32, -1)
    # (a) teacher-forced: oracle inputs, rounded to bf16 once
    tf = mod.run(cl(ho), tb, None, skip=cl(skip_o) if is_out else None)
    # (b) free-running engine
    he = mod.run(he, tb, None, skip=skip_e)
    e_tf, e_fr = err(back(tf), ref), err(back(he), ref)
    kinds = "+".join(type(m).__name__ for m in mod)
    rows.append((name, tuple(ref.shape[1:]), kinds, e_tf, e_fr))
    ho = ref
    if not is_out and name != "middle_block":
        hs_o.append(ho)
        hs_e.append(he)
ops.stats_end(dev)
# head
ref_logits = O.conv(O.silu(O.group_norm(ho, sd["out.0.weight"], sd["out.0.bias"], 1e-5)), sd["out.2.weight"], sd["out.2.bias"], padding=1)
full = u.forward_cl(cl(x), bias_row)
tf_head_in = cl(ho)
from jointimagegeneration_amd.blocks import norm_conv, packed_conv, _k3  # noqa: E402
pw, pb = packed_conv(u.out[2], tf_head_in.Cpad)
ops.stats_begin(dev)
tf_head = norm_conv(tf_head_in, u.out[0], True, pw, pb, K, k=_k3(u.out[2].weight), out_f32=True)
ops.stats_end(dev)
rows.append(("out (GN+SiLU+conv -> logits)", tuple(ref_logits.shape[1:]), "head", err(back(tf_head), ref_logits), err(back(full), ref_logits)))
print(f"CCDM UNet @{R}^3, bf16 engine vs fp32 oracle; errors relative to the rms of the oracle activation")
print(f"{'block':34s} {'shape':22s} {'added by the block (rms / max)':32s} {'accumulated (rms / max)':28s} layers")
for name, shp, kinds, (a, am), (b, bm) in rows:
    print(f"{name:34s} {str(shp):22s} {a:10.3e} / {am:10.3e}        {b:10.3e} / {bm:10.3e}    {kinds}")
lo = back(full)
p_e, p_o = torch.softmax(lo, 1), torch.softmax(ref_logits, 1)
print(f"logits: abs err rms {float((lo - ref_logits).pow(2).mean().sqrt()):.3e} max {float((lo - ref_logits).abs().max()):.3e} (logit rms {float(ref_logits.pow(2).mean().sqrt()):.3f}); "
      f"probabilities: max abs err {float((p_e - p_o).abs().max()):.3e}")
