"""Upper bound of folding GroupNorm statistics in the producing conv (conv1 -> GN2 of every ResBlock): time the captured latent-UNet
forward with GN2 applied by the plain gn_apply kernel from a DUMMY scale / shift table (numerics are garbage: timing only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from jointimagegeneration_amd import ops, blocks as B
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
orig = B.norm_conv
dummy = {}
def patched(h, norm, act, weight, bias, cout, src2=None, **kw):
    if kw.get("residual") is not None and src2 is None and not ops.conv_fuses_prologue(h, cout, **kw):      # conv2 of a ResBlock: its norm is GN2
        key = h.Cpad
        if key not in dummy:
            dummy[key] = (torch.ones(1, h.Cpad, device=dev), torch.zeros(1, h.Cpad, device=dev))
        a = ops.groupnorm_apply(h, dummy[key][0], dummy[key][1], act)
        return ops.conv(a, weight, bias, cout, **kw)
    return orig(h, norm, act, weight, bias, cout, src2=src2, **kw)
def run(tag):
    u.forward_cl(x, row); torch.cuda.synchronize()
    g = ops.capture_graph(lambda: u.forward_cl(x, row))
    for _ in range(5): g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"{tag}: {e0.elapsed_time(e1) * 10:.1f} us per forward", flush=True)
for rnd in range(2):
    B.norm_conv = orig; run("production")
    B.norm_conv = patched; run("GN2 by plain gn_apply (dummy table)")
