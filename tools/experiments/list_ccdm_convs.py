import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from jointimagegeneration_amd import ops
from jointimagegeneration_amd.ops import CL
from jointimagegeneration_amd.unet import create_unet_openai
# shape-only listing on CPU tensors is impossible (ops need the GPU); emulate by wrapping ops.conv to record and raise no error
calls=[]
orig=ops.conv
def rec(src1, weight, bias, cout, k=(1,3,3), stride=1, pad=1, upsample=False, src2=None, **kw):
    calls.append((tuple(src1.t.shape), None if src2 is None else src2.t.shape[-1], cout, k, stride, upsample, 'pro' if kw.get('prologue') is not None else ''))
    return orig(src1, weight, bias, cout, k=k, stride=stride, pad=pad, upsample=upsample, src2=src2, **kw)
ops.conv=rec
import jointimagegeneration_amd.blocks as B
B.ops.conv=rec
torch.set_grad_enabled(False)
dev=torch.device("cuda:0")
from jointimagegeneration_amd.synth import randomize_parameters
u = create_unet_openai(image_size=128, in_channels=15, out_channels=14, num_res_blocks=2, cond_encoded_shape=None, dims=3,
                       base_channels=64, channel_mult=[1, 2, 2, 4, 5], attention_resolutions=[32, 16, 8], num_heads=1,
                       num_head_channels=32, softmax_output=True).eval()
randomize_parameters(u, 1024, "ccdm."); u=u.to(dev)
R=128
x = CL(torch.zeros(1, R, R, R, 32, dtype=torch.bfloat16, device=dev), 15); x.t[...,0]=1
row = u.time_bias_rows(torch.tensor([17.0], device=dev))
u.forward_cl(x,row); torch.cuda.synchronize()
import collections
agg=collections.Counter(calls)
for k,v in sorted(agg.items(), key=lambda kv:(-kv[0][0][1],kv[0][3])):
    sh,c2,cout,kk,st,up,pro=k
    print(f"{v:3d} x  in {sh[1:4]} C={sh[4]}+{c2 or 0} -> {cout:4d}  k={kk} stride={st} up={up} {pro}")
