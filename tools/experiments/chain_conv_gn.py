"""Captured chains of the latent UNet's two main kernels on one level (C channels, HW x HW, batch 1), us per launch:
   (a) 100 x conv3x3 alone (x -> y -> x ...), (b) 100 x GroupNorm*SiLU apply from accumulators alone, (c) 100 x [conv3x3 with the
   statistics epilogue -> GroupNorm*SiLU apply from ITS accumulators] as in a ResBlock.  (c) minus (a) minus (b) is what the
   alternation itself costs (cross-XCD hand-off of the activations, accumulator atomics and their read-back).
   python tools/experiments/chain_conv_gn.py C HW"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from jointimagegeneration_amd import ops
from jointimagegeneration_amd.ops import CL
torch.set_grad_enabled(False)
Cc, HW = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
x0 = CL(torch.randn(1, 1, HW, HW, Cc, device=dev).bfloat16(), Cc)
w = ops.pack_conv_weight((torch.randn(Cc, Cc, 3, 3, device=dev) / math.sqrt(Cc * 9))[:, :, None], Cc)
b = ops.pad_bias(torch.zeros(Cc, device=dev), Cc, dev)
gam, bet = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
L = 100
def timeit(fn, per):
    fn(); torch.cuda.synchronize()
    g = ops.capture_graph(fn)
    for _ in range(3): g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (20 * per) * 1e3
def conv_chain(stats):
    def f():
        ops.stats_begin(dev)
        h = x0
        for _ in range(L): h = ops.conv(h, w, b, Cc, k=(1, 3, 3), want_stats=stats)
        ops.stats_end(dev)
    return f
def gn_chain():
    ops.stats_begin(dev)
    h0 = ops.conv(x0, w, b, Cc, k=(1, 3, 3), want_stats=True)       # a tensor WITH accumulators
    ops.stats_end(dev)
    def f():
        h = h0
        for _ in range(L):
            y = ops.groupnorm_apply_acc(h, gam, bet, 1e-5, True)
            y.acc = h0.acc                                            # keep reading the same accumulators
            h = y
    return f
def pair_chain():
    def f():
        ops.stats_begin(dev)
        h = x0
        for _ in range(L):
            h = ops.conv(h, w, b, Cc, k=(1, 3, 3), want_stats=True)
            h = ops.groupnorm_apply_acc(h, gam, bet, 1e-5, True)
        ops.stats_end(dev)
    return f
ops.GN_ACC = False
tc0 = timeit(conv_chain(False), L)              # no statistics epilogue at all (no accumulator atomics)
ops.GN_ACC = True
tc1 = timeit(conv_chain(True), L)
try:
    tg = timeit(gn_chain(), L)
except Exception as e:
    tg = float("nan"); print("gn chain failed:", e)
tp = timeit(pair_chain(), L)
print(f"C={Cc} @{HW}^2: conv alone {tc0:.2f} us, conv + statistics epilogue {tc1:.2f} us, GroupNorm apply alone {tg:.2f} us, conv -> GroupNorm pair {tp:.2f} us per PAIR (sum of the parts {tc1 + tg:.2f})")
