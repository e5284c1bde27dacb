"""Phase stamps of gn_apply_acc_kernel (diagnostic build: hipcc -DGG_GN_STAMPS of gg_norm.hip linked into tools/experiments/ab/libG.so):
   python tools/experiments/probe_gn_stamps.py C HW
(lean kernel; GG_OLD_KERNEL=1 with a -DGG_GN_APPLY_ACC_LEAN=0 build for the table kernel)  Prints over all blocks the s_memrealtime (10 ns ticks) of thread 0 at: entry (arguments pinned), loads issued, first accumulator landed,
group sums folded, statistics done, table written, stored -- relative to the first block's entry."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from jointimagegeneration_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ab", "libG.so")
import ctypes as C
import numpy as np
import torch
from jointimagegeneration_amd import ops
torch.set_grad_enabled(False)
Cc, HW = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
x = ops.CL(torch.randn(1, 1, HW, HW, Cc, device=dev).bfloat16(), Cc)
acc = torch.zeros(1, Cc, 2, dtype=torch.int64, device=dev)
xf = x.t.float().reshape(-1, Cc)
acc[0, :, 0] = (xf.sum(0).double() * 2 ** 28).round().long()
acc[0, :, 1] = ((xf * xf).sum(0).double() * 2 ** 20).round().long()
gam, bet = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
out = torch.empty_like(x.t)
lib = _lib.load()
raw = C.CDLL(_lib.LIB_PATH)
st = torch.cuda.current_stream().cuda_stream
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    e0.record()
    _lib.check(lib.gg_groupnorm_apply_acc(x.t.data_ptr(), Cc, acc.data_ptr(), None, 0, None, 1, HW * HW, Cc, gam.data_ptr(), bet.data_ptr(),
                                          C.c_float(1e-5), 1, out.data_ptr(), st), "gn")
    e1.record()
torch.cuda.synchronize()
nb = min(4096, (HW * HW * Cc // 8 + 255) // 256)
buf = np.zeros(nb * 8, dtype=np.uint64)
assert raw.gg_gn_stamps_read(buf.ctypes.data_as(C.c_void_p), nb * 8) == 0
t = buf.reshape(nb, 8).astype(np.float64)
t0 = t[:, 0].min()
names = ["entry (args pinned)", "loads issued, sums parked", "after barrier 1", "group statistics done", "after barrier 2", "stored"]
if os.environ.get("GG_OLD_KERNEL"): names = ["entry (args pinned)", "loads issued", "first acc landed", "group sums folded", "statistics done", "table written", "stored"]
print(f"gn_apply_acc C={Cc} @{HW}^2: {nb} blocks, event time {e0.elapsed_time(e1) * 1e3:.1f} us; us after the first block's entry:")
for i, nm in enumerate(names):
    v = (t[:, i] - t0) / 100.0
    print(f"  {nm:20s} min {v.min():6.2f}  mean {v.mean():6.2f}  max {v.max():6.2f}")
