"""Same-box A/B of library builds on the first-stage AE decode and the cond-stage encode at 512^2 (hipGraph replay):
   python tools/experiments/ab_ae.py tools/experiments/ab/libA.so   (one process per library)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from jointimagegeneration_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from jointimagegeneration_amd import ops
from jointimagegeneration_amd.ops import CL
from jointimagegeneration_amd.synth import randomize_parameters
from jointimagegeneration_amd.ldm import AutoencoderKL
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
res = []
for name, inch, ch in (("first_stage", 1, 128), ("cond_stage", 2, 96)):
    a = AutoencoderKL(ddconfig=dict(double_z=True, z_channels=4, resolution=512, in_channels=inch, out_ch=inch, ch=ch, ch_mult=[1, 2, 4, 4],
                                    num_res_blocks=2, dropout=0.0, dims=2, attn_resolutions=[16, 8]), embed_dim=4, dims=2).eval()
    randomize_parameters(a, 1024, name + "."); a = a.to(dev)
    if inch == 1:
        z = CL(torch.randn(1, 1, 64, 64, 32, device=dev).bfloat16(), 4)
        fn = lambda: a.decode_cl(z)
    else:
        x = CL(torch.randn(1, 1, 512, 512, 32, device=dev).bfloat16(), 2)
        fn = lambda: a.encode_moments_cl(x)
    fn(); torch.cuda.synchronize()
    g = ops.capture_graph(fn)
    for _ in range(3): g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): g.replay()
    e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 20)
print(f"{os.path.basename(sys.argv[1])}: decode {res[0]:.3f} ms, cond-encode {res[1]:.3f} ms", flush=True)
