"""Phase stamps of the team halo conv (diagnostic build -DGG_H3_STAMPS: tools/experiments/build_variant.sh h3stamps "-DGG_H3_STAMPS" gg_conv gg_conv_halo3).
Workgroup 0, wave 0 (team A) and wave 4 (team B).  S-phase slots: 2 iv = arrival at barrier iv, 2 iv + 1 = release; T-phase: 20 start, 21 asm end, 22 released.
python tools/experiments/probe_halo3_stamps.py Cin Cout S [pro]"""
import sys, os
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import ctypes as C
import torch
from jointimagegeneration_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "experiments", "ab", os.environ.get("H3_LIB", "libh3stamps.so"))
from jointimagegeneration_amd import ops
from jointimagegeneration_amd._lib import ConvDesc, GG_BF16
torch.set_grad_enabled(False)
Cin, Cout, S = [int(a) for a in sys.argv[1:4]]
pro = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dev = torch.device("cuda:0")
x = torch.randn(1, S, S, S, Cin, device=dev).bfloat16()
w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) / (Cin * 27) ** 0.5
pw = ops.pack_conv_weight(w, Cin)
pb = ops.pad_bias(None, Cout, dev)
sc, sh = ops.groupnorm_stats(ops.CL(x, Cin), torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev), 1e-5)
out = torch.empty(1, S, S, S, ops.pad32(Cout), dtype=torch.bfloat16, device=dev)
ws = torch.zeros(2 * 64 * 32, dtype=torch.int64, device=dev)
lib = _lib.load()
d = ConvDesc()
d.N, d.D, d.H, d.W = 1, S, S, S
d.C1, d.C2, d.Cout, d.Cout_pad = Cin, 0, Cout, ops.pad32(Cout)
d.kd, d.kh, d.kw, d.stride, d.pad, d.upsample = 3, 3, 3, 1, 1, 0
d.Do, d.Ho, d.Wo = S, S, S
d.out_dtype = GG_BF16
d.prologue_act = pro
d.path_hint = 7
d.src1, d.weight, d.bias, d.out = x.data_ptr(), pw.data_ptr(), pb.data_ptr(), out.data_ptr()
if pro:
    d.gn_scale, d.gn_shift = sc.data_ptr(), sh.data_ptr()
d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 8
for _ in range(3):
    ws.zero_()
    _lib.check(lib.gg_conv_forward(C.byref(d), torch.cuda.current_stream().cuda_stream), "conv")
torch.cuda.synchronize()
t = ws.cpu().view(2, 64, 32)
t0 = int(t[t > 0].min())
print(f"conv {Cin}->{Cout} {S}^3 pro={pro}: cycles (s_memtime, 100 MHz-independent shader clock) relative to the first stamp")
for team in range(2):
    print("team", "AB"[team])
    for ph in range(64):
        r = t[team, ph]
        if int(r.max()) == 0:
            continue
        if int(r[20]) > 0:
            print(f"  phase {ph:2d} T: start {int(r[20]) - t0:8d}  asm {int(r[21] - r[20]):6d}  end-barrier wait {int(r[22] - r[21]):6d}   P: work {int(r[25] - r[24]):6d} barrier {int(r[26] - r[25]):6d}")
        else:
            work = [int(r[0]) - (int(t[team, ph - 1][22]) if ph and int(t[team, ph - 1][22]) else int(r[0]))] + [int(r[2 * i] - r[2 * i - 1]) for i in range(1, 10)]
            wait = [int(r[2 * i + 1] - r[2 * i]) for i in range(10)]
            print(f"  phase {ph:2d} S: start {int(r[0]) - t0:8d}  total {int(r[19] - r[0]):6d}  work/iv {work}  wait/iv {wait}   P: work {int(r[25] - r[24]):6d} barrier {int(r[26] - r[25]):6d}")
