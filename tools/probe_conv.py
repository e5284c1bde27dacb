"""Micro-benchmark of one conv shape: python tools/probe_conv.py N C_in C_out H W [k]   (2-D, stride 1)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jointimagegeneration_amd import ops
torch.set_grad_enabled(False)
N, Cin, Cout, H, W = [int(a) for a in sys.argv[1:6]]
k = int(sys.argv[6]) if len(sys.argv) > 6 else 3
dev = torch.device("cuda:0")
x = ops.CL(torch.randn(N, 1, H, W, Cin, device=dev).bfloat16(), Cin)
w = torch.randn(Cout, Cin, k, k, device=dev) / (Cin * k * k) ** 0.5
pw = ops.pack_conv_weight(w, Cin)
pb = ops.pad_bias(None, Cout, dev)
f = lambda: ops.conv(x, pw, pb, Cout, k=(1, k, k), pad=k // 2)
if os.environ.get("PROBE_PRO"):      # GroupNorm*SiLU in front of the conv: fused prologue when the kernel takes it, else apply + conv
    gamma, beta = torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev)
    sc, sh = ops.groupnorm_stats(x, gamma, beta, 1e-5)
    if ops.conv_fuses_prologue(x, Cout, k=(1, k, k), pad=k // 2):
        f = lambda: ops.conv(x, pw, pb, Cout, k=(1, k, k), pad=k // 2, prologue=(sc, sh))
    else:
        f = lambda: ops.conv(ops.groupnorm_apply(x, sc, sh, True), pw, pb, Cout, k=(1, k, k), pad=k // 2)
if os.environ.get("PROBE_ACC"): ops.stats_begin(dev)      # convs emit GroupNorm sums into the arena
f(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(50): f()
g.replay(); torch.cuda.synchronize()
t0 = time.time()
for _ in range(5): g.replay()
torch.cuda.synchronize()
t = (time.time() - t0) / 250
gf = 2.0 * N * H * W * Cout * Cin * k * k / 1e9
print(f"conv N={N} {Cin}->{Cout} {H}x{W} k={k}: {t*1e6:.1f} us/conv ({gf/t/1e3:.1f} TFLOP/s) pro={os.environ.get('PROBE_PRO')}")
