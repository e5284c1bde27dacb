"""A captured chain of 200 dependent gn_apply_acc launches (x -> y -> x ...), us per launch: what the kernel costs when its neighbours are
itself (compare with its ~4.8 us inside the latent-UNet forward and with the 1.7-3.1 us of the trivial kernels of ubench_*.hip).
   python tools/experiments/chain_gn.py C HW"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import torch
from jointimagegeneration_amd import _lib, ops
torch.set_grad_enabled(False)
Cc, HW = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
x = torch.randn(1, 1, HW, HW, Cc, device=dev).bfloat16(); y = torch.empty_like(x)
acc = torch.zeros(1, Cc, 2, dtype=torch.int64, device=dev)
xf = x.float().reshape(-1, Cc)
acc[0, :, 0] = (xf.sum(0).double() * 2 ** 28).round().long(); acc[0, :, 1] = ((xf * xf).sum(0).double() * 2 ** 20).round().long()
gam, bet = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
lib = _lib.load()
def chain():
    st = torch.cuda.current_stream().cuda_stream
    a, b = x, y
    for _ in range(200):
        _lib.check(lib.gg_groupnorm_apply_acc(a.data_ptr(), Cc, acc.data_ptr(), None, 0, None, 1, HW * HW, Cc, gam.data_ptr(), bet.data_ptr(),
                                              C.c_float(1e-5), 1, b.data_ptr(), st), "gn")
        a, b = b, a
chain(); torch.cuda.synchronize()
g = ops.capture_graph(chain)
for _ in range(3): g.replay()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): g.replay()
e1.record(); torch.cuda.synchronize()
print(f"gn_apply_acc C={Cc} @{HW}^2 chained: {e0.elapsed_time(e1) / (20 * 200) * 1e3:.2f} us per launch")
