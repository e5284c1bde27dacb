"""What do COLD weights cost a latent-UNet conv?  A captured chain of 128 dependent 3x3 convs (C -> C at HW x HW, batch 1) rotating
through NW distinct packed weight tensors: 1 (always L2-warm), a few dozen (beyond the 8 x 4 MB of L2, inside the 256 MB MALL), a few
hundred (beyond the MALL: every launch streams its weights from HBM, as every layer of the real forward does).  us per conv.
   python tools/experiments/chain_conv_weights.py C HW"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from jointimagegeneration_amd import ops
from jointimagegeneration_amd.ops import CL
torch.set_grad_enabled(False)
Cc, HW = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
x0 = CL(torch.randn(1, 1, HW, HW, Cc, device=dev).bfloat16(), Cc)
b = ops.pad_bias(torch.zeros(Cc, device=dev), Cc, dev)
wf = (torch.randn(Cc, Cc, 3, 3, device=dev) / math.sqrt(Cc * 9))[:, :, None]
mb = Cc * Cc * 9 * 2 / 1e6
L = 128
for NW in (1, 4, 16, 64, int(700 / mb) + 1):
    ws = [ops.pack_conv_weight(wf, Cc) for _ in range(NW)]
    def f():
        h = x0
        for i in range(L): h = ops.conv(h, ws[i % NW], b, Cc, k=(1, 3, 3))
    f(); torch.cuda.synchronize()
    g = ops.capture_graph(f)
    reps = max(3, (NW + L - 1) // L * 3)
    for _ in range(reps): g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"C={Cc} @{HW}^2, {NW:4d} weight sets of {mb:.1f} MB ({NW * mb:7.1f} MB in rotation): {e0.elapsed_time(e1) / (20 * L) * 1e3:.2f} us per conv", flush=True)
    del ws, g
