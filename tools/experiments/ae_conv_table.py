"""Every conv launch of one AE decode (64^2 -> 512^2) and one cond-encode (512^2 -> 64^2) with shape, duration (HIP events, eager) and rate:
python tools/experiments/ae_conv_table.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from jointimagegeneration_amd import ops
from jointimagegeneration_amd.ops import CL
from jointimagegeneration_amd.synth import randomize_parameters
from jointimagegeneration_amd.ldm import AutoencoderKL
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
recs = []
real_conv = ops.conv
def timed(src1, weight, bias, cout, k=(1, 3, 3), stride=1, pad=1, upsample=False, src2=None, **kw):
    halo = ops.conv_runs_halo_tile(src1, cout, k=k, stride=stride, pad=pad, upsample=upsample, src2=src2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = real_conv(src1, weight, bias, cout, k=k, stride=stride, pad=pad, upsample=upsample, src2=src2, **kw)
    e1.record()
    cin = src1.C + (src2.C if src2 is not None else 0)
    M = out.t.shape[0] * out.t.shape[1] * out.t.shape[2] * out.t.shape[3]
    recs.append((e0, e1, tuple(src1.t.shape[2:4]), cin, cout, k, stride, pad, upsample, "prologue" if (kw.get("prologue") is not None or kw.get("prologue_acc") is not None) else "", "res" if kw.get("residual") is not None else "", halo, 2.0 * M * cout * cin * k[0] * k[1] * k[2]))
    return out
for name, inch, ch, dec in (("first_stage", 1, 128, True), ("cond_stage", 2, 96, False)):
    a = AutoencoderKL(ddconfig=dict(double_z=True, z_channels=4, resolution=512, in_channels=inch, out_ch=inch, ch=ch, ch_mult=[1, 2, 4, 4], num_res_blocks=2,
                                    dropout=0.0, dims=2, attn_resolutions=[16, 8]), embed_dim=4, dims=2).eval()
    randomize_parameters(a, 1024, name + "."); a = a.to(dev)
    x = CL(torch.randn(1, 1, 64, 64, 32, device=dev).bfloat16(), 4) if dec else CL(torch.randn(1, 1, 512, 512, 32, device=dev).bfloat16(), 2)
    f = (lambda: a.decode_cl(x)) if dec else (lambda: a.encode_moments_cl(x))
    f(); f()
    ops.conv = timed
    try:
        recs.clear()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
    finally:
        ops.conv = real_conv
    tot = 0.0
    print(f"== {'decode 64^2 -> 512^2 (first stage, ch 128)' if dec else 'cond-encode 512^2 -> 64^2 (cond stage, ch 96)'}")
    for (t0, t1, sp, cin, cout, k, st, pd, up, pro, res, halo, fl) in recs:
        ms = t0.elapsed_time(t1); tot += ms
        print(f"{str(sp):12s} {cin:4d}->{cout:4d} k{k[1]}{k[2]} s{st} p{pd} {'up' if up else '  '} {pro:8s} {res:3s} {'halo' if halo else '    '} {ms*1e3:8.1f} us {fl/ms/1e9:7.0f} TF/s")
    print(f"convs {tot:.2f} ms of {e0.elapsed_time(e1):.2f} ms (eager, with event pairs)")
