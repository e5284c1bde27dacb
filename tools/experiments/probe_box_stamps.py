"""Phase stamps of conv_box2d_kernel (diagnostic build: hipcc -DGG_BOX_STAMPS of gg_conv.hip + gg_conv_box.hip linked into
tools/experiments/ab/libS.so):  python tools/experiments/probe_box_stamps.py C1 C2 Cout HW k [prologue]
Prints, over all workgroups, the s_memrealtime (10 ns ticks) of: entry, box DMAs issued, first weight trips issued, box landed (and prologue done), k-loop done,
all waves done, combine visible, end -- relative to the first workgroup's entry."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from jointimagegeneration_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ab", os.environ.get("GG_STAMP_LIB", "libS.so"))
import ctypes as C
import torch
from jointimagegeneration_amd import ops
from jointimagegeneration_amd._lib import ConvDesc, GG_BF16
torch.set_grad_enabled(False)
C1, C2, Cout, HW, k = [int(a) for a in sys.argv[1:6]]
pro = int(sys.argv[6]) if len(sys.argv) > 6 else 0
dev = torch.device("cuda:0")
x1 = torch.randn(1, 1, HW, HW, C1, device=dev).bfloat16()
x2 = torch.randn(1, 1, HW, HW, C2, device=dev).bfloat16() if C2 else None
w = torch.randn(Cout, C1 + C2, k, k, device=dev) / ((C1 + C2) * k * k) ** 0.5
pw = ops.pack_conv_weight(w[:, :, None], C1 + C2)
pb = ops.pad_bias(None, Cout, dev)
out = torch.empty(1, 1, HW, HW, ops.pad32(Cout), dtype=torch.bfloat16, device=dev)
res = torch.randn_like(out)
ws = torch.zeros(1024 * 2 * 16, dtype=torch.int64, device=dev)
sc = torch.ones(1, C1 + C2, device=dev); sh = torch.zeros(1, C1 + C2, device=dev)
lib = _lib.load()
d = ConvDesc()
d.N, d.D, d.H, d.W = 1, 1, HW, HW
d.C1, d.C2, d.Cout, d.Cout_pad = C1, C2, Cout, ops.pad32(Cout)
d.kd, d.kh, d.kw, d.stride, d.pad, d.upsample = 1, k, k, 1, (1 if k == 3 else 0), 0
d.Do, d.Ho, d.Wo = 1, HW, HW
d.out_dtype = GG_BF16
d.prologue_act = pro
d.path_hint = 98
d.src1, d.weight, d.bias, d.out = x1.data_ptr(), pw.data_ptr(), pb.data_ptr(), out.data_ptr()
if x2 is not None: d.src2 = x2.data_ptr()
d.residual = res.data_ptr()
if pro: d.gn_scale, d.gn_shift = sc.data_ptr(), sh.data_ptr()
d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 8
st = torch.cuda.current_stream().cuda_stream
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    ws.zero_()
    e0.record()
    _lib.check(lib.gg_conv_forward(C.byref(d), st), "conv")
    e1.record()
torch.cuda.synchronize()
t = ws.cpu().view(1024, 2, 16)
nb = int((t[:, 0, 0] != 0).sum())
t = t[:nb].double()
t0 = t[:, 0, 0].min()
names = ["entry", "box DMAs issued", "weights issued", "box landed (+pro)", "k-loop done", "all waves done", "combine visible", "end", "before DMA loop", "1 DMA issued", "4 DMAs issued"]
print(f"box conv {C1}+{C2}->{Cout} @{HW}^2 k{k} pro={pro}: {nb} workgroups, event time {e0.elapsed_time(e1) * 1e3:.1f} us (incl. launch gap); us after first entry:")
for wv in (0, 1):
    for i, nm in enumerate(names):
        v = (t[:, wv, i] - t0) / 100.0
        print(f"  wave {'0' if wv == 0 else '7'} {nm:18s} min {v.min():6.2f}  mean {v.mean():6.2f}  max {v.max():6.2f}")
