"""Latent-UNet forward (N=1, 64x64, hipGraph) against ops.GN_ACC_MIN_ELEMS: python tools/experiments/probe_gn_acc_min.py"""
import sys, os, time
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
vals = [int(v) for v in sys.argv[1:]] or [1 << 18, 1 << 17, 1 << 16, 1 << 15, 1 << 14]
outs = {}
for rnd in range(2):
    for v in vals:
        ops.GN_ACC_MIN_ELEMS = v
        u.forward_cl(x, row); torch.cuda.synchronize()
        g = ops.capture_graph(lambda: u.forward_cl(x, row))
        for _ in range(3): g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): g.replay()
        e1.record(); torch.cuda.synchronize()
        print(f"GN_ACC_MIN_ELEMS={v}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per forward", flush=True)
