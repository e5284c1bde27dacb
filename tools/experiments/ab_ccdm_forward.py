"""Same-box A/B of library builds on the CCDM UNet forward (128^3, hipGraph replay):
   python tools/experiments/ab_ccdm_forward.py tools/experiments/ab/libA.so   (one process per library)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from jointimagegeneration_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from jointimagegeneration_amd import ops
from jointimagegeneration_amd.ops import CL
from jointimagegeneration_amd.synth import randomize_parameters
from jointimagegeneration_amd.unet import create_unet_openai
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
u = create_unet_openai(image_size=128, in_channels=15, out_channels=14, num_res_blocks=2, cond_encoded_shape=None, dims=3,
                       base_channels=64, channel_mult=[1, 2, 2, 4, 5], attention_resolutions=[32, 16, 8], num_heads=1,
                       num_head_channels=32, softmax_output=True).eval()
randomize_parameters(u, 1024, "ccdm."); u = u.to(dev)
x = CL(torch.zeros(1, 128, 128, 128, 32, dtype=torch.bfloat16, device=dev), 15); x.t[..., 0] = 1
row = u.time_bias_rows(torch.tensor([17.0], device=dev))
u.forward_cl(x, row); torch.cuda.synchronize()
g = ops.capture_graph(lambda: u.forward_cl(x, row))
for _ in range(2): g.replay()
ts = []
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 10)
print(f"{os.path.basename(sys.argv[1])}: " + " / ".join(f"{t:.3f}" for t in ts) + " ms per CCDM forward", flush=True)
