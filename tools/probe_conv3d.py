"""Micro-benchmark of one 3-D conv: python tools/probe_conv3d.py Cin Cout S [reps]   (N=1, 3x3x3, stride 1, S^3)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jointimagegeneration_amd import ops
torch.set_grad_enabled(False)
Cin, Cout, S = [int(a) for a in sys.argv[1:4]]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device("cuda:0")
x = ops.CL(torch.randn(1, S, S, S, Cin, device=dev).bfloat16(), Cin)
w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) / (Cin * 27) ** 0.5
pw = ops.pack_conv_weight(w, Cin)
pb = ops.pad_bias(None, Cout, dev)
f = lambda: ops.conv(x, pw, pb, Cout, k=(3, 3, 3))
if os.environ.get("PROBE_PRO"):      # with the fused GroupNorm*SiLU prologue (the way the CCDM ResBlocks call it)
    sc, sh = ops.groupnorm_stats(x, torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev), 1e-5)
    f = lambda: ops.conv(x, pw, pb, Cout, k=(3, 3, 3), prologue=(sc, sh))
f(); torch.cuda.synchronize()
t0 = time.time()
for _ in range(reps): f()
torch.cuda.synchronize()
t = (time.time() - t0) / reps
gf = 2.0 * S ** 3 * Cout * Cin * 27 / 1e9
print(f"conv3d {Cin}->{Cout} {S}^3: {t*1e6:.1f} us ({gf/t/1e3:.1f} TFLOP/s)")
