"""Viability test: prefetch the weights of the low-resolution convs into the memory-side cache from a SIDE stream of the captured
latent-UNet forward (fork edges only; one join at the end).  A dry run records, per ops.conv call, its packed weight tensor; in the
captured run, when the main stream reaches conv i, the side stream (after an event of the main stream) reads the weights of the convs
i + AHEAD .. i + AHEAD + every - 1 that are >= MIN_MB (torch int32 sum = a plain streaming read).
   python tools/experiments/ab_prefetch_side_stream.py [ahead] [every] [min_mb]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from jointimagegeneration_amd import ops
import jointimagegeneration_amd.blocks as B
from jointimagegeneration_amd.ops import CL
from jointimagegeneration_amd.synth import randomize_parameters
from jointimagegeneration_amd.unet import UNetModel
torch.set_grad_enabled(False)
AHEAD, EVERY, MIN_MB = int(sys.argv[1]) if len(sys.argv) > 1 else 3, int(sys.argv[2]) if len(sys.argv) > 2 else 4, float(sys.argv[3]) if len(sys.argv) > 3 else 4.0
dev = torch.device("cuda:0")
u = UNetModel(dims=2, image_size=512, in_channels=8, out_channels=4, model_channels=160, attention_resolutions=[8, 4, 2],
              num_res_blocks=2, channel_mult=[1, 2, 4, 4, 5], num_head_channels=32).eval()
randomize_parameters(u, 1024, "ldm."); u = u.to(dev)
x = CL(torch.randn(1, 1, 64, 64, 32, device=dev).bfloat16(), 8)
row = u.time_bias_rows(torch.full((1,), 981.0, device=dev))
orig = ops.conv
calls = []
def rec(src1, weight, *a, **k):
    calls.append(weight)
    return orig(src1, weight, *a, **k)
ops.conv = rec; B.ops.conv = rec
u.forward_cl(x, row); torch.cuda.synchronize()
weights = list(calls)
big = [i for i, w in enumerate(weights) if w.numel() * w.element_size() >= MIN_MB * 1e6]
print(f"{len(weights)} conv calls, {len(big)} with weights >= {MIN_MB} MB ({sum(weights[i].numel() * 2 for i in big) / 1e6:.0f} MB)")
side = torch.cuda.Stream()
sink = torch.zeros(64, dtype=torch.int64, device=dev)
def make(prefetch):
    state = {"i": 0}
    def hooked(src1, weight, *a, **k):
        i = state["i"]; state["i"] += 1
        if prefetch and i % EVERY == 0:
            todo = [j for j in range(i + AHEAD, i + AHEAD + EVERY) if j in bigset]
            if todo:
                ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    side.wait_event(ev)
                    for n, j in enumerate(todo): sink[n % 64] = weights[j].view(torch.int32).sum()
        return orig(src1, weight, *a, **k)
    def fwd():
        state["i"] = 0
        ops.conv = hooked; B.ops.conv = hooked
        u.forward_cl(x, row)
        if prefetch: torch.cuda.current_stream().wait_stream(side)
    return fwd
bigset = set(big)
for name, pf in (("no prefetch", False), ("side-stream prefetch", True), ("no prefetch", False), ("side-stream prefetch", True)):
    f = make(pf)
    f(); torch.cuda.synchronize()
    g = ops.capture_graph(f)
    for _ in range(5): g.replay()
    ts = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): g.replay()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 100 * 1e3)
    print(f"{name:22s} (ahead {AHEAD}, every {EVERY}): " + " / ".join(f"{t:.1f}" for t in ts) + " us per forward", flush=True)
