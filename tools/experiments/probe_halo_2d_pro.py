"""2-D halo conv with / without the fused GroupNorm * SiLU prologue at the autoencoder's shapes, beside the cost of a separate apply pass:
python tools/experiments/probe_halo_2d_pro.py   -- decides whether 'apply as its own pass + prologue-free conv' could beat the fused form."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
if os.environ.get("H3_LIB"):
    from jointimagegeneration_amd import _lib
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ab", os.environ["H3_LIB"])
from jointimagegeneration_amd import ops
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
def timeit(f, n=20):
    f(); torch.cuda.synchronize()
    for rnd in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / n * 1e3
    return t
for (Cin, Cout, S, dims3) in [(128, 128, 512, 0), (256, 256, 256, 0), (512, 512, 128, 0), (512, 512, 64, 0), (96, 96, 512, 0), (192, 192, 256, 0), (384, 384, 128, 0), (384, 384, 64, 0),
        (64, 64, 128, 1), (256, 256, 32, 1), (128, 128, 64, 1), (512, 512, 16, 1), (256, 128, 64, 1), (512, 256, 32, 1)]:
    shp = (1, S, S, S) if dims3 else (1, 1, S, S)
    g = torch.Generator(device="cpu").manual_seed(3)
    x = ops.CL(torch.randn(shp + (Cin,), generator=g).to(dev).bfloat16(), Cin)
    k = (3, 3, 3) if dims3 else (1, 3, 3)
    w = torch.randn((Cout, Cin) + (k if dims3 else k[1:]), generator=g).to(dev) / (Cin * 9) ** 0.5
    pw = ops.pack_conv_weight(w, Cin)
    bias = torch.zeros(1, ops.pad32(Cout), device=dev)
    gamma, beta = torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev)
    prol = ops.groupnorm_stats(x, gamma, beta, 1e-5)
    t1 = timeit(lambda: ops.conv(x, pw, bias, Cout, k=k, bias_per_sample=True, prologue=prol))
    t0 = timeit(lambda: ops.conv(x, pw, bias, Cout, k=k, bias_per_sample=True))
    ta = timeit(lambda: ops.groupnorm_apply(x, prol[0], prol[1], True))
    gf = 2.0 * shp[1] * shp[2] * shp[3] * Cout * Cin * (27 if dims3 else 9) / 1e9
    print(f"{'3' if dims3 else '2'}-D {Cin}->{Cout} @{S}: fused prologue {t1:.1f} us ({gf / t1 * 1e3:.0f} TF/s); no prologue {t0:.1f} us ({gf / t0 * 1e3:.0f} TF/s); "
          f"separate apply pass {ta:.1f} us", flush=True)
