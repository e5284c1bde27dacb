"""Same-process A/B of ops-level switches on the captured latent-UNet forward (N=1, 8x64x64): python tools/experiments/ab_flags_ldm.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from jointimagegeneration_amd import ops
from jointimagegeneration_amd.ops import CL
from jointimagegeneration_amd.synth import randomize_parameters
from jointimagegeneration_amd.unet import UNetModel
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
u = UNetModel(dims=2, image_size=512, in_channels=8, out_channels=4, model_channels=160, attention_resolutions=[8, 4, 2],
              num_res_blocks=2, channel_mult=[1, 2, 4, 4, 5], num_head_channels=32).eval()
randomize_parameters(u, 1024, "ldm."); u = u.to(dev)
x = CL(torch.randn(1, 1, 64, 64, 32, device=dev).bfloat16(), 8)
row = u.time_bias_rows(torch.full((1,), 981.0, device=dev))
nl = [0]
real = ops.check


def run(tag):
    out = u.forward_cl(x, row); torch.cuda.synchronize()
    g = ops.capture_graph(lambda: u.forward_cl(x, row))
    for _ in range(5): g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"{tag:44s}: {e0.elapsed_time(e1) * 5:.1f} us per forward", flush=True)
    return out.t.float().clone()


arms = [("baseline (no skip K-concat, launch GN)", dict(SKIP_KCONCAT=False, PROLOGUE_FROM_ACC=False)),
        ("accumulator prologue only", dict(SKIP_KCONCAT=False, PROLOGUE_FROM_ACC=True)),
        ("skip K-concat only", dict(SKIP_KCONCAT=True, PROLOGUE_FROM_ACC=False)),
        ("both (production)", dict(SKIP_KCONCAT=True, PROLOGUE_FROM_ACC=True))]
res = {}
for rnd in range(2):
    for tag, kw in arms:
        saved = {k: getattr(ops, k) for k in kw}
        for k, v in kw.items():
            setattr(ops, k, v)
        res[tag] = run(tag)
        for k, v in saved.items():
            setattr(ops, k, v)
a, b = res[arms[0][0]], res[arms[3][0]]
print(f"eps difference baseline vs production: max {float((a - b).abs().max()):.3e} (eps max {float(a.abs().max()):.3f})")
