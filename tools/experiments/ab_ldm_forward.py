"""Same-box A/B of library builds on the latent-UNet forward (N=1, 64x64, hipGraph replay):
   python tools/experiments/ab_ldm_forward.py tools/experiments/ab/libA.so   (one process per library; alternate them in one gpurun call)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from jointimagegeneration_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from jointimagegeneration_amd import ops
from jointimagegeneration_amd.ops import CL
from jointimagegeneration_amd.synth import randomize_parameters
from jointimagegeneration_amd.unet import UNetModel
torch.set_grad_enabled(False)
ops.TINY_IMAGE_POSITIONS = int(os.environ.get("GG_TINY", ops.TINY_IMAGE_POSITIONS))      # A/B of the SiLU-norm fold at the deepest levels
dev = torch.device("cuda:0")
u = UNetModel(dims=2, image_size=512, in_channels=8, out_channels=4, model_channels=160, attention_resolutions=[8, 4, 2],
              num_res_blocks=2, channel_mult=[1, 2, 4, 4, 5], num_head_channels=32).eval()
randomize_parameters(u, 1024, "ldm."); u = u.to(dev)
NB, RR = int(os.environ.get("GG_N", 1)), int(os.environ.get("GG_R", 64))          # batch / latent size (default: the C5 slice, N = 1 @64x64)
x = CL(torch.randn(NB, 1, RR, RR, 32, device=dev).bfloat16(), 8)
row = u.time_bias_rows(torch.full((NB,), 981.0, device=dev))
u.forward_cl(x, row); torch.cuda.synchronize()
g = ops.capture_graph(lambda: u.forward_cl(x, row))
for _ in range(5): g.replay()
ts = []
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): g.replay()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 100 * 1e3)
print(f"{os.path.basename(sys.argv[1])} N={NB} @{RR}: " + " / ".join(f"{t:.1f}" for t in ts) + " us per forward", flush=True)
