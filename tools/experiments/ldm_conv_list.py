"""Shapes of every conv launch of one latent-UNet forward (N = 1 @64x64), in launch order, for joining with the kernel order of a
captured replay (tools/experiments/ldm_graph_timeline.py):  python tools/experiments/ldm_conv_list.py > list.txt   (no GPU timing here)"""
import os, sys
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
u.forward_cl(x, row)
real = ops.conv
def spy(src1, weight, bias, cout, k=(1, 3, 3), stride=1, pad=1, upsample=False, src2=None, **kw):
    cin = src1.C + (src2.C if src2 is not None else 0)
    tags = [t for t, on in (("up", upsample), (f"s{stride}", stride != 1), ("2src", src2 is not None), ("res", kw.get("residual") is not None),
                            ("pro", kw.get("prologue") is not None), ("pro_acc", kw.get("prologue_acc") is not None), ("skip", kw.get("skip") is not None),
                            ("ddim", kw.get("ddim") is not None)) if on]
    print(f"{tuple(src1.t.shape[2:4])} {cin}->{cout} k{k[1]}{k[2]} {' '.join(tags)}")
    return real(src1, weight, bias, cout, k=k, stride=stride, pad=pad, upsample=upsample, src2=src2, **kw)
ops.conv = spy
try:
    u.forward_cl(x, row)
finally:
    ops.conv = real
