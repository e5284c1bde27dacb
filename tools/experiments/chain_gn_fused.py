"""A captured chain of 200 dependent gn_fused_small launches (x -> y -> x ...), us per launch (compare with ~4.7 us inside the forward).
   python tools/experiments/chain_gn_fused.py C HW"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from jointimagegeneration_amd import ops
from jointimagegeneration_amd.ops import CL
torch.set_grad_enabled(False)
Cc, HW = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
x = CL(torch.randn(1, 1, HW, HW, Cc, device=dev).bfloat16(), Cc)
gam, bet = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
assert ops.groupnorm_fused_ok(x)
def chain():
    h = x
    for _ in range(200): h = ops.groupnorm_fused(h, gam, bet, 1e-5, False)
chain(); torch.cuda.synchronize()
g = ops.capture_graph(chain)
for _ in range(3): g.replay()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): g.replay()
e1.record(); torch.cuda.synchronize()
print(f"gn_fused_small C={Cc} @{HW}^2 chained: {e0.elapsed_time(e1) / (20 * 200) * 1e3:.2f} us per launch")
