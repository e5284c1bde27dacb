"""Per-slice wall time of the LDM stage at the C5 shapes (N=1, 512^2, 50 DDIM steps): python tools/probe_slices.py [slices]
First call warms (eager slice + capture), second call is timed."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jointimagegeneration_amd.pipeline import GuideGenPipeline, build_ccdm, build_ldm
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
pipe = GuideGenPipeline(build_ccdm(14, 250, 1024, dev), build_ldm(1024, dev), ddim_steps=50)
labels = torch.zeros(1, 128, 128, 128, dtype=torch.int32, device=dev)
labels[:, :, 32:96, 32:96] = 3
pipe.sample_ct(labels, 256, 512, 7, max_slices=3)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.time()
    pipe.sample_ct(labels, 256, 512, 8 + rep, max_slices=n)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f"{n} slices: {dt / n * 1e3:.2f} ms per slice (x256 = {dt / n * 256:.2f} s per volume)", flush=True)
print({k: round(v, 4) for k, v in pipe.time_slice_stages(1, 512).items()})
