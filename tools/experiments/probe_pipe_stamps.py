"""In-kernel phase stamps of the pipelined halo conv (diagnostic build -DGG_PIPE_ABLATIONS, path_hint 32): block 0, item 2.
python tools/probe_pipe_stamps.py Cin Cout S"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from jointimagegeneration_amd import ops, _lib
from jointimagegeneration_amd._lib import ConvDesc, GG_BF16
torch.set_grad_enabled(False)
Cin, Cout, S = [int(a) for a in sys.argv[1:4]]
dev = torch.device("cuda:0")
x = torch.randn(1, S, S, S, Cin, device=dev).bfloat16()
w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) / (Cin * 27) ** 0.5
pw = ops.pack_conv_weight(w, Cin)
pb = ops.pad_bias(None, Cout, dev)
sc, sh = ops.groupnorm_stats(ops.CL(x, Cin), torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev), 1e-5)
out = torch.empty(1, S, S, S, ops.pad32(Cout), dtype=torch.bfloat16, device=dev)
ws = torch.zeros(8 * 9 * 8, dtype=torch.int64, device=dev)
lib = _lib.load()
d = ConvDesc()
d.N, d.D, d.H, d.W = 1, S, S, S
d.C1, d.C2, d.Cout, d.Cout_pad = Cin, 0, Cout, ops.pad32(Cout)
d.kd, d.kh, d.kw, d.stride, d.pad, d.upsample = 3, 3, 3, 1, 1, 0
d.Do, d.Ho, d.Wo = S, S, S
d.out_dtype = GG_BF16
d.prologue_act = 1 if os.environ.get("PROBE_PRO", "1") == "1" else 0
d.path_hint = 32
d.src1, d.weight, d.bias, d.out = x.data_ptr(), pw.data_ptr(), pb.data_ptr(), out.data_ptr()
d.gn_scale, d.gn_shift = sc.data_ptr(), sh.data_ptr()
d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 8
for _ in range(3):
    _lib.check(lib.gg_conv_forward(C.byref(d), torch.cuda.current_stream().cuda_stream), "conv")
torch.cuda.synchronize()
t = ws.cpu().view(8, 9, 8)
t0 = int(t[:, 0, 0].min())
names = ["start", "dma", "tr_pre", "mfma", "tr_post", "wait", "barrier"]
print("cycles since the item's first stamp; columns: " + " ".join(names))
for g in range(9):
    for wv in (0, 1, 4, 5):
        row = [int(t[wv, g, k]) - t0 for k in range(7)]
        print(f"g={g} wave={wv}: " + " ".join(f"{v:7d}" for v in row) + "   | deltas " + " ".join(f"{row[k+1]-row[k]:6d}" for k in range(6)))
print("item total cycles:", int(t[:, 8, 6].max()) - t0)
