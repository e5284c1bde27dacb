"""Same-box A/B of the halo conv epilogue (bias / residual loads ahead of the stores): python tools/experiments/probe_halo_epi.py   (H3_LIB=libprev_halo.so = before)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
if os.environ.get("H3_LIB"):
    from jointimagegeneration_amd import _lib
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ab", os.environ["H3_LIB"])
from jointimagegeneration_amd import ops
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
for (dims, Cin, Cout, S, pro, res) in [(3, 64, 64, (128, 128, 128), 1, 1), (3, 64, 64, (128, 128, 128), 1, 0), (3, 192, 64, (128, 128, 128), 1, 1), (3, 128, 128, (64, 64, 64), 1, 1),
                                        (3, 256, 256, (32, 32, 32), 1, 1), (2, 128, 128, (1, 512, 512), 1, 1), (2, 256, 256, (1, 256, 256), 1, 1)]:
    g = torch.Generator(device="cpu").manual_seed(3)
    x = ops.CL(torch.randn((1,) + S + (Cin,), generator=g).to(dev).bfloat16(), Cin)
    k = (3, 3, 3) if dims == 3 else (1, 3, 3)
    w = torch.randn((Cout, Cin) + k[3 - dims:], generator=g).to(dev) / (Cin * 27) ** 0.5
    pw = ops.pack_conv_weight(w, Cin)
    bias = torch.zeros(1, ops.pad32(Cout), device=dev); bias[:, :Cout] = torch.randn(1, Cout, generator=g).to(dev)
    r = ops.CL(torch.randn((1,) + S + (ops.pad32(Cout),), generator=g).to(dev).bfloat16(), Cout) if res else None
    gamma, beta = torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev)
    prol = ops.groupnorm_stats(x, gamma, beta, 1e-5) if pro else None
    f = lambda: ops.conv(x, pw, bias, Cout, k=k, residual=r, bias_per_sample=True, prologue=prol)
    y = f(); torch.cuda.synchronize()
    for rnd in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20 * 1e3
    gf = 2.0 * S[0] * S[1] * S[2] * Cout * Cin * (27 if dims == 3 else 9) / 1e9
    print(f"{dims}-D {Cin}->{Cout} @{S} pro={pro} res={res}: {t:.1f} us ({gf / t * 1e3:.0f} TF/s)  checksum {float(y.t.float().sum()):.6e}", flush=True)
