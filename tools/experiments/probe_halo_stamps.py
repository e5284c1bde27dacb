import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ctypes as C
import torch
from jointimagegeneration_amd import ops, _lib
from jointimagegeneration_amd._lib import ConvDesc, GG_BF16
torch.set_grad_enabled(False)
Cin, Cout, S = [int(a) for a in sys.argv[1:4]]
pro = int(os.environ.get("PROBE_PRO", "1"))
dev = torch.device("cuda:0")
x = torch.randn(1, S, S, S, Cin, device=dev).bfloat16()
w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) / (Cin * 27) ** 0.5
pw = ops.pack_conv_weight(w, Cin)
pb = ops.pad_bias(None, Cout, dev)
sc, sh = ops.groupnorm_stats(ops.CL(x, Cin), torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev), 1e-5)
out = torch.empty(1, S, S, S, ops.pad32(Cout), dtype=torch.bfloat16, device=dev)
ws = torch.zeros(4 * 8 * 8, dtype=torch.int64, device=dev)
lib = _lib.load()
d = ConvDesc()
d.N, d.D, d.H, d.W = 1, S, S, S
d.C1, d.C2, d.Cout, d.Cout_pad = Cin, 0, Cout, ops.pad32(Cout)
d.kd, d.kh, d.kw, d.stride, d.pad, d.upsample = 3, 3, 3, 1, 1, 0
d.Do, d.Ho, d.Wo = S, S, S
d.out_dtype = GG_BF16
d.prologue_act = pro
d.path_hint = 99
d.src1, d.weight, d.bias, d.out = x.data_ptr(), pw.data_ptr(), pb.data_ptr(), out.data_ptr()
d.gn_scale, d.gn_shift = sc.data_ptr(), sh.data_ptr()
d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 8
for _ in range(3):
    _lib.check(lib.gg_conv_forward(C.byref(d), torch.cuda.current_stream().cuda_stream), "conv")
torch.cuda.synchronize()
t = ws.cpu().view(4, 8, 8)
nch = Cin // 32
t0 = int(t[:, 0, 0].min())
print(f"conv {Cin}->{Cout} {S}^3 pro={pro}: block 2049, per chunk: [start, staged, after barrier] and end of taps; cycles")
for wv in range(4):
    for c in range(nch):
        row = [int(t[wv, c, k]) - t0 for k in range(3)]
        nxt = int(t[wv, c + 1, 0]) - t0 if c + 1 < nch else int(t[wv, c, 3]) - t0
        print(f"wave {wv} chunk {c}: start {row[0]:7d} staging {row[1]-row[0]:6d} barrier {row[2]-row[1]:6d} taps {nxt-row[2]:6d}")
