"""A/B of the 3-D halo conv with 512-position boxes (2 workgroups per CU) against 256-position boxes (path_hint 1 / 4: 2 / 3 per CU).
python tools/experiments/probe_halo_halfbox.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from jointimagegeneration_amd import ops
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
for (Cin, Cout, S, pro) in [(64, 64, 128, 1), (192, 64, 128, 1), (128, 128, 64, 1), (64, 64, 128, 0), (256, 256, 32, 1), (32, 64, 128, 1)]:
    x = ops.CL(torch.randn(1, S, S, S, Cin, device=dev).bfloat16(), Cin)
    w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) / (Cin * 27) ** 0.5
    pw = ops.pack_conv_weight(w, Cin); pb = ops.pad_bias(None, Cout, dev)
    kw = {}
    if pro:
        sc, sh = ops.groupnorm_stats(x, torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev), 1e-5)
        kw = dict(prologue=(sc, sh))
    outs = {}
    for rnd in range(3):
        for hint in (1, 6):
            ops.PATH_HINT = hint
            y = ops.conv(x, pw, pb, Cout, k=(3, 3, 3), **kw); torch.cuda.synchronize()
            outs[hint] = y.t
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): ops.conv(x, pw, pb, Cout, k=(3, 3, 3), **kw)
            e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / 30 * 1e3
            gf = 2.0 * S ** 3 * Cout * Cin * 27 / 1e9
            print(f"{Cin}->{Cout} @{S}^3 pro={pro} hint={hint}: {t:.1f} us ({gf/t*1e3:.0f} TF/s)", flush=True)
    print("   bit-identical:", bool(torch.equal(outs[1], outs[6])), flush=True)
