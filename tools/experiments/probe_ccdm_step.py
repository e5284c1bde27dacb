"""Captured CCDM reverse step @128^3 (UNet forward + softmax / posterior / draw), ms per step, with the reverse step fused into the head conv
(ops.FUSE_POSTERIOR, default) and as its own launch:   python tools/experiments/probe_ccdm_step.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from jointimagegeneration_amd import ops
from jointimagegeneration_amd.pipeline import build_ccdm
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
ccdm = build_ccdm(14, 250, 1024, dev)


def chain(steps, seed=12):
    g = torch.Generator(device=dev).manual_seed(seed)
    x_T = torch.randint(0, 14, (1, 128, 128, 128), generator=g, device=dev, dtype=torch.int32)
    ccdm.philox_seed = seed
    cond = torch.zeros((1, 1, 128, 128, 128), device=dev)
    torch.cuda.synchronize(); t0 = time.time()
    lab, _ = ccdm.sample_labels(x_T, cond, 10000 + steps)
    torch.cuda.synchronize()
    return time.time() - t0, lab


ref = None
for rnd in range(2):
    for fuse in (True, False):
        ops.FUSE_POSTERIOR = fuse
        chain(6)
        t10 = min(chain(10)[0] for _ in range(2))
        t30, lab = min((chain(30) for _ in range(2)), key=lambda r: r[0])
        same = "" if ref is None else f"  labels equal to the first run: {bool(torch.equal(ref, lab))}"
        ref = lab if ref is None else ref
        print(f"fused reverse step {fuse}: {(t30 - t10) / 20 * 1e3:.3f} ms per captured step (last_step_fused {getattr(ccdm, 'last_step_fused', None)}){same}", flush=True)
