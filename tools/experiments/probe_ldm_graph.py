"""Latent-UNet forward N=1 @64x64 as a captured hipGraph, replayed: for a rocprofv3 --kernel-trace pass that shows the kernel
durations and the gaps BETWEEN kernels inside a replay (tools/experiments/ldm_graph_timeline.py reads the database).
   rocprofv3 --kernel-trace -d out -o ldmg -- python3 tools/experiments/probe_ldm_graph.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
if os.environ.get("H3_LIB"):
    from jointimagegeneration_amd import _lib
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ab", os.environ["H3_LIB"])
from jointimagegeneration_amd.unet import UNetModel
from jointimagegeneration_amd.synth import randomize_parameters
from jointimagegeneration_amd.ops import CL
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
u = UNetModel(dims=2, image_size=512, in_channels=8, out_channels=4, model_channels=160, attention_resolutions=[8, 4, 2],
              num_res_blocks=2, channel_mult=[1, 2, 4, 4, 5], num_head_channels=32).eval()
randomize_parameters(u, 1024, "ldm."); u = u.to(dev)
x = CL(torch.randn(1, 1, 64, 64, 32, device=dev).bfloat16(), 8)
row = u.time_bias_rows(torch.full((1,), 981.0, device=dev))
for _ in range(2):
    u.forward_cl(x, row)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    u.forward_cl(x, row)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
print(f"graph replay: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
