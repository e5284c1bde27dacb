"""Which phase are the two co-resident workgroups of a CU in?  (diagnostic build of gg_conv_halo.hip with stamps, path_hint 99)"""
import sys, os, collections
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ctypes as C
import torch
from jointimagegeneration_amd import ops, _lib
from jointimagegeneration_amd._lib import ConvDesc, GG_BF16
torch.set_grad_enabled(False)
Cin, Cout, S = 64, 64, 128
dev = torch.device("cuda:0")
x = torch.randn(1, S, S, S, Cin, device=dev).bfloat16()
w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) / (Cin * 27) ** 0.5
pw = ops.pack_conv_weight(w, Cin); pb = ops.pad_bias(None, Cout, dev)
sc, sh = ops.groupnorm_stats(ops.CL(x, Cin), torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev), 1e-5)
out = torch.empty(1, S, S, S, 64, dtype=torch.bfloat16, device=dev)
ws = torch.zeros(512 * 12, dtype=torch.int64, device=dev)
lib = _lib.load()
d = ConvDesc()
d.N, d.D, d.H, d.W = 1, S, S, S
d.C1, d.C2, d.Cout, d.Cout_pad = Cin, 0, Cout, 64
d.kd, d.kh, d.kw, d.stride, d.pad, d.upsample = 3, 3, 3, 1, 1, 0
d.Do, d.Ho, d.Wo = S, S, S
d.out_dtype = GG_BF16; d.prologue_act = 1; d.path_hint = 99
d.src1, d.weight, d.bias, d.out = x.data_ptr(), pw.data_ptr(), pb.data_ptr(), out.data_ptr()
d.gn_scale, d.gn_shift = sc.data_ptr(), sh.data_ptr()
d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 8
for _ in range(3):
    _lib.check(lib.gg_conv_forward(C.byref(d), torch.cuda.current_stream().cuda_stream), "conv")
torch.cuda.synchronize()
t = ws.cpu().view(512, 12)
byc = collections.defaultdict(list)
for b in range(512):
    hw, xcc = int(t[b, 0]), int(t[b, 1])
    cu = (hw >> 8) & 0xF; sh_ = (hw >> 12) & 1; se = (hw >> 13) & 7; wave_id = hw & 0xF; simd = (hw >> 4) & 3; tg = (hw >> 16) & 0xF
    byc[(xcc & 0xF, se, sh_, cu)].append((int(t[b, 2]), b, wave_id, simd, tg, [int(v) for v in t[b, 2:11]]))
t0 = int(t[:, 2].min())
n = 0
for key, lst in sorted(byc.items()):
    lst.sort()
    if n < 6:
        print("CU", key)
        for st, b, wid, simd, tg, ts in lst:
            r = [v - t0 for v in ts]      # 100 MHz ticks -> x10 ns
            print(f"   block {b+1024} wave_slot {wid} simd {simd} tg {tg}: chunk0 start {r[0]*10:6d} ns staged {r[1]*10:6d} taps {r[2]*10:6d} | chunk1 start {r[4]*10:6d} staged {r[5]*10:6d} taps {r[6]*10:6d} end {r[7]*10:6d}")
    n += 1
print("CUs seen", len(byc), "blocks per CU", collections.Counter(len(v) for v in byc.values()))
