import sys, os, math
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from jointimagegeneration_amd import ops
from oracle import nets as O
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
def bf(x): return x.to(torch.bfloat16).float()
for (N, Cin, Cout, sp) in ((1, 32, 32, (16, 16)), (1, 32, 32, (8, 8)), (1, 160, 160, (16, 16))):
    g = torch.Generator().manual_seed(1)
    x = torch.randn((N, Cin) + sp, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    ref = O.conv(bf(x), bf(w), None, padding=1)
    xcl = ops.to_cl(x.to(dev))
    out = ops.conv(xcl, ops.pack_conv_weight(w.to(dev), xcl.Cpad), ops.pad_bias(None, Cout, dev), Cout, k=(1, 3, 3))
    got = ops.from_cl(out, 2).cpu()
    err = (got - ref).abs()
    print(sp, Cin, Cout, "max err", float(err.max()), "ref max", float(ref.abs().max()), "nan", int(torch.isnan(got).sum()))
    e = err[0].amax(0)      # per position
    print((e > 0.05).int())
    print("per-channel bad:", (err[0].amax((1, 2)) > 0.05).nonzero().flatten().tolist()[:40])
