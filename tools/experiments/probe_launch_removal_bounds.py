"""Upper bounds (timing only, numerics are garbage) of removing launches from the captured latent-UNet forward at batch 1:
  no_attn_gn : the GroupNorm launch in front of every AttentionBlock's qkv conv dropped
  no_skip    : the 1x1 skip_connection conv of every channel-changing ResBlock dropped (no residual)
  no_gn      : every GroupNorm launch dropped
Each arm = what a perfect fusion of those launches into their neighbours would buy."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn as nn
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
orig_norm_conv, orig_gn_silu, orig_res_run, orig_attn_run = B.norm_conv, B.gn_silu, B.ResBlock.run, B.AttentionBlock.run
flags = dict(no_attn_gn=False, no_skip=False, no_gn=False)
nlaunch = [0]


def norm_conv(h, norm, act, weight, bias, cout, src2=None, **kw):
    if flags["no_gn"] or (flags["no_attn_gn"] and not act and kw.get("k") == (1, 1, 1)):
        if src2 is not None:
            return ops.conv(h, weight, bias, cout, src2=src2, **kw)
        return ops.conv(h, weight, bias, cout, **kw)
    return orig_norm_conv(h, norm, act, weight, bias, cout, src2=src2, **kw)


def res_run(self, h, tbias, src2=None):
    if not flags["no_skip"] or isinstance(self.skip_connection, nn.Identity):
        return orig_res_run(self, h, tbias, src2)
    c1, c2 = self.in_layers[2], self.out_layers[3]
    k = B._k3(c1.weight)
    cin_pad = h.Cpad + (src2.Cpad if src2 is not None else 0)
    pw1, _ = B.packed_conv(c1, cin_pad)
    h1 = B.norm_conv(h, self.in_layers[0], True, pw1, tbias, self.out_channels, src2=src2, k=k, bias_per_sample=True)
    pw2, pb2 = B.packed_conv(c2, h1.Cpad)
    return B.norm_conv(h1, self.out_layers[0], True, pw2, pb2, self.out_channels, k=k)


B.norm_conv = norm_conv
B.ResBlock.run = res_run
import jointimagegeneration_amd.unet as U
U.norm_conv = norm_conv


def run(tag):
    u.forward_cl(x, row); torch.cuda.synchronize()
    g = ops.capture_graph(lambda: u.forward_cl(x, row))
    for _ in range(5): g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"{tag:40s}: {e0.elapsed_time(e1) * 10:.1f} us per forward", flush=True)


for rnd in range(2):
    for name in ("production", "no_attn_gn", "no_skip", "no_attn_gn+no_skip", "no_gn", "no_gn+no_skip"):
        for k in flags:
            flags[k] = k in name.split("+")
        run(name)
