"""GroupNorm micro-benchmark: python tools/probe_gn.py C H W   (N=1): stats + apply launches vs apply from conv-epilogue accumulators"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jointimagegeneration_amd import ops
torch.set_grad_enabled(False)
C, H, W = [int(a) for a in sys.argv[1:4]]
dev = torch.device("cuda:0")
x = ops.CL(torch.randn(1, 1, H, W, C, device=dev).bfloat16(), C)
w = torch.randn(C, C, 1, 1, device=dev) / C ** 0.5
pw = ops.pack_conv_weight(w, C)
ops.stats_begin(dev)
y = ops.conv(x, pw, ops.pad_bias(None, C, dev), C, k=(1, 1, 1), pad=0)      # leaves y.acc behind
ops.stats_end(dev)
gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
def old():
    sc, sh = ops.groupnorm_stats(y, gamma, beta, 1e-5)
    return ops.groupnorm_apply(y, sc, sh, True)
def only_apply(sc_sh=ops.groupnorm_stats(y, gamma, beta, 1e-5)):
    return ops.groupnorm_apply(y, sc_sh[0], sc_sh[1], True)
def new():
    return ops.groupnorm_apply_acc(y, gamma, beta, 1e-5, True)
def fused():
    return ops.groupnorm_fused(y, gamma, beta, 1e-5, True)
cases = [("stats+apply", old), ("apply only", only_apply)] + ([("fused one-launch", fused)] if ops.groupnorm_fused_ok(y) else []) + ([("apply_acc", new)] if y.acc is not None else [])
for name, f in cases:
    f(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(50): f()
    g.replay(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    print(f"GN C={C} {H}x{W} {name}: {(time.time() - t0) / 250 * 1e6:.2f} us")
