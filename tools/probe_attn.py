"""Attention micro-benchmark: python tools/probe_attn.py T heads d   (N=1, self-attention, head-major q|k|v rows as in the UNet)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jointimagegeneration_amd import ops
torch.set_grad_enabled(False)
T, H, D = [int(a) for a in sys.argv[1:4]]
dev = torch.device("cuda:0")
qkv = torch.randn(1, T, 3 * H * D, device=dev).bfloat16()
out = torch.empty(1, T, H * D, device=dev, dtype=torch.bfloat16)
ld = 3 * H * D
f = lambda: ops.attention(qkv, qkv, qkv, out, 1, H, D, T, T, (ld, 3 * D), (ld, 3 * D), (ld, 3 * D), (H * D, D), D ** -0.5, 0, D, 2 * D)
f(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(50): f()
g.replay(); torch.cuda.synchronize()
t0 = time.time()
for _ in range(5): g.replay()
torch.cuda.synchronize()
t = (time.time() - t0) / 250
print(f"attention T={T} heads={H} d={D}: {t*1e6:.1f} us ({4.0*T*T*D*H/t/1e12:.1f} TFLOP/s)")
