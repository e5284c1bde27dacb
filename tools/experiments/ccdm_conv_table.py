"""Every conv launch of one CCDM UNet forward @128^3 with its shape, duration (HIP events, eager) and rate:
python tools/experiments/ccdm_conv_table.py   -- where the forward's time is, launch by launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from jointimagegeneration_amd import ops
from jointimagegeneration_amd.ops import CL
from jointimagegeneration_amd.pipeline import build_ccdm
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
unet = build_ccdm(14, 250, 1024, dev).unet
x = CL(torch.zeros(1, 128, 128, 128, 32, dtype=torch.bfloat16, device=dev), 15)
x.t[..., 0] = 1
row = unet.time_bias_rows(torch.tensor([17.0], device=dev))
unet.forward_cl(x, row)
recs = []
real_conv, real = ops.conv, {}
def timed(src1, weight, bias, cout, k=(1, 3, 3), stride=1, pad=1, upsample=False, src2=None, **kw):
    halo = ops.conv_runs_halo_tile(src1, cout, k=k, stride=stride, pad=pad, upsample=upsample, src2=src2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = real_conv(src1, weight, bias, cout, k=k, stride=stride, pad=pad, upsample=upsample, src2=src2, **kw)
    e1.record()
    cin = src1.C + (src2.C if src2 is not None else 0)
    M = out.t.shape[0] * out.t.shape[1] * out.t.shape[2] * out.t.shape[3]
    recs.append((e0, e1, tuple(src1.t.shape[1:4]), cin, cout, k, stride, upsample, "prologue" if (kw.get("prologue") is not None or kw.get("prologue_acc") is not None) else "", "res" if kw.get("residual") is not None else "", halo, 2.0 * M * cout * cin * k[0] * k[1] * k[2]))
    return out
ops.conv = timed
try:
    for _ in range(2):
        recs.clear()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); unet.forward_cl(x, row); e1.record(); torch.cuda.synchronize()
finally:
    ops.conv = real_conv
tot = 0.0
for (a, b, sp, cin, cout, k, st, up, pro, res, halo, fl) in recs:
    ms = a.elapsed_time(b); tot += ms
    print(f"{str(sp):16s} {cin:4d}->{cout:4d} k{k[0]}{k[1]}{k[2]} s{st} {'up' if up else '  '} {pro:8s} {res:3s} {'halo' if halo else '    '} {ms*1e3:8.1f} us {fl/ms/1e9:7.0f} TF/s")
print(f"convs {tot:.2f} ms of the forward's {e0.elapsed_time(e1):.2f} ms (eager, with event pairs)")
