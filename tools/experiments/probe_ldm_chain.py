"""One captured DDIM chain (50 steps, N = 1 @64x64) of the latent UNet as the pipeline runs it, ms per replay; `ops.STATS_CHAIN_SLOTS` was the
experiment "one zero fill of the GroupNorm accumulators per chain" (README: slower, not kept; without that attribute both modes are the
product):   python tools/experiments/probe_ldm_chain.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from jointimagegeneration_amd import ops
from jointimagegeneration_amd.ldm import DDIMSampler
from jointimagegeneration_amd.pipeline import build_ldm
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
ldm = build_ldm(1024, dev)
res = {}
for rnd in range(2):
    for flag in (True, False):
        ops.STATS_CHAIN_SLOTS = flag
        sampler = DDIMSampler(ldm)
        sampler.make_schedule(50, ddim_eta=0.0, verbose=False)
        st = sampler.prepare_state(1, 4, (64, 64), dev, 4)
        g = torch.Generator(device=dev).manual_seed(5)
        x_T = torch.randn((1, 1, 64, 64, 4), generator=g, device=dev)
        outs = []
        for it in range(8):                       # warm (eager), capture, replays
            st["x"].copy_(x_T); st["unet_in"][..., :4].copy_(x_T)
            torch.cuda.synchronize(); t0 = time.time()
            sampler.run_steps(st, None, 0.0, None)
            torch.cuda.synchronize()
            outs.append((time.time() - t0, st["x"].clone()))
        best = min(t for t, _ in outs[3:])
        same = all(torch.equal(outs[0][1], o) for _, o in outs[1:])
        res.setdefault(flag, outs[-1][1])
        print(f"chain slots {flag}: {best * 1e3:.3f} ms per 50-step chain = {best * 1e3 / 50:.4f} ms per step; eager == replays: {same}; "
              f"equal to the other mode: {bool(torch.equal(res[True], outs[-1][1])) if True in res else None}", flush=True)
