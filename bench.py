#!/usr/bin/env python3
"""GuideGen sampling benchmark on MI355X (contract in the task statement; metric from BASELINE.json).

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one complete GuideGen volume per GPU (config C5 of BASELINE.json, one independent volume per rank):
  CCDM 3-D categorical UNet, 128^3 mask, 14 classes, 250 reverse steps  ->  stage glue  ->
  LDM autoregressive CT, 256 slices of 512x512: per slice cond-stage encode + 50 DDIM steps (latent 4x64x64) + AE decode.
value = sampled voxels/s = (N * K * 69 206 016 voxels) / wall time of the K steps (max over ranks), synthetic inputs and
random-init weights of the reference architectures (no checkpoints or data exist offline), inputs resident in HBM.
Volumes are independent => weak scaling, no collective on the data path.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

VOXELS_PER_VOLUME = 128 ** 3 + 512 * 512 * 256          # 69 206 016
MFMA_PEAK_TFLOPS = 2500.0                                # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
CCDM_UNET_GFLOP_128 = 12720.7


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--ccdm-steps", type=int, default=250, help="CCDM reverse steps (250 = params_eval.yml time_steps)")
    ap.add_argument("--slices", type=int, default=256)
    ap.add_argument("--volumes-per-gpu", type=int, default=1, help="independent volumes sampled concurrently per GPU (BASELINE C5 = 1)")
    ap.add_argument("--max-slices", type=int, default=None, help="DEV ONLY: truncate the slice loop (marks the line partial)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def conv_roofline(pipe, device):
    """Dominant kernel = the 3-D halo-tile implicit-GEMM conv (`conv_halo_kernel<1,NT,UP>`) of the CCDM UNet
    (3x3x3 convs are 99.3 % of the UNet's FLOPs; the halo kernel runs all of them at the 128^3..32^3 levels).
    One eager 128^3 UNet forward with a HIP event pair around EVERY conv launch, recorded on the stream the kernels are
    launched on; achieved = sum(algorithmic FLOPs of the halo launches) / sum(their durations).  The rocprofv3 average of
    the same kernel name (profiles/) must agree with avg_launch_ms.  traffic: PMC passes parsed by tools/pmc_parse.py."""
    from jointimagegeneration_amd import ops
    from jointimagegeneration_amd.ops import CL
    unet = pipe.ccdm.unet
    K = pipe.ccdm.diffusion.num_classes
    x = CL(torch.zeros(1, 128, 128, 128, 32, dtype=torch.bfloat16, device=device), K + 1)
    x.t[..., 0] = 1
    row = unet.time_bias_rows(torch.tensor([17.0], device=device))
    unet.forward_cl(x, row)                                              # warm
    records = []
    real_conv = ops.conv

    def timed_conv(src1, weight, bias, cout, k=(1, 3, 3), stride=1, pad=1, upsample=False, src2=None, **kw):
        halo = ops.conv_fuses_prologue(src1, cout, k=k, stride=stride, pad=pad, upsample=upsample, src2=src2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = real_conv(src1, weight, bias, cout, k=k, stride=stride, pad=pad, upsample=upsample, src2=src2, **kw)
        e1.record()
        cin = src1.C + (src2.C if src2 is not None else 0)
        M = out.t.shape[0] * out.t.shape[1] * out.t.shape[2] * out.t.shape[3]
        records.append((e0, e1, 2.0 * M * cout * cin * k[0] * k[1] * k[2], k, halo))
        return out

    ops.conv = timed_conv
    try:
        unet.forward_cl(x, row)
        torch.cuda.synchronize()
    finally:
        ops.conv = real_conv
    halo = [(e0.elapsed_time(e1) * 1e-3, fl) for e0, e1, fl, k, h in records if h and k[0] == 3]
    k27 = [(e0.elapsed_time(e1) * 1e-3, fl) for e0, e1, fl, k, h in records if k[0] * k[1] * k[2] == 27]
    t, fl = sum(a for a, _ in halo), sum(b for _, b in halo)
    t27, fl27 = sum(a for a, _ in k27), sum(b for _, b in k27)
    ach = fl / t / 1e12
    out = {"bound": "mfma", "kernel": "conv_halo_kernel<3-D> (3x3x3 implicit GEMM, halo tile) in one CCDM UNet forward @128^3",
           "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4),
           "traffic": None, "launches": len(halo), "avg_launch_ms": round(t / len(halo) * 1e3, 4), "flops_per_launch": round(fl / len(halo)),
           "all_3x3x3_convs": {"launches": len(k27), "achieved": round(fl27 / t27 / 1e12, 1), "flops_per_forward": fl27}}
    pmc = os.path.join(ROOT, "profiles", "r01_pmc_conv3d.json")
    if os.path.exists(pmc):
        j = json.load(open(pmc))
        out["traffic"] = j["traffic_bytes_per_launch"]
        out["traffic_source"] = "profiles/r01_pmc_conv3d.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2 gfx950 correction)"
    return out


def cpu_baseline():
    """Oracle (CPU fp32 restatement, validated against the reference in the build container) timed on the host cores on a
    bounded sample of the same workload and extrapolated linearly: one CCDM UNet forward at 64^3 (x8 -> 128^3), one LDM
    UNet forward (N=1, 64^2), one AE decode and one cond-stage encode at 512^2."""
    from jointimagegeneration_amd.pipeline import CCDM_PARAMS, LDM_UNET, ae_config
    from jointimagegeneration_amd.config import instantiate_from_config
    from jointimagegeneration_amd.synth import randomize_parameters
    from jointimagegeneration_amd.unet import create_unet_openai
    from oracle import nets as O
    # the GPU box gives one GPU a 16-core share of the host; os.cpu_count() reports the whole host there
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("GG_CPU_CORES", "16")))
    torch.set_num_threads(cores)
    with torch.no_grad():
        u = create_unet_openai(image_size=128, in_channels=15, out_channels=14, num_res_blocks=2, cond_encoded_shape=None, dims=3, **CCDM_PARAMS)
        randomize_parameters(u, 1024, "ccdm.")
        sd = {k: v for k, v in u.state_dict().items()}
        x = torch.zeros(1, 15, 64, 64, 64); x[:, 0] = 1
        t0 = time.time(); O.unet_forward(sd, x, torch.tensor([17.0]), model_channels=64, head_channels=32, softmax_out=True); t_ccdm64 = time.time() - t0
        u2 = instantiate_from_config(LDM_UNET); randomize_parameters(u2, 1024, "ldm.")
        sd2 = dict(u2.state_dict())
        x2 = torch.randn(1, 8, 64, 64)
        O.unet_forward(sd2, x2, torch.tensor([981]), model_channels=160, head_channels=32)
        t0 = time.time(); O.unet_forward(sd2, x2, torch.tensor([981]), model_channels=160, head_channels=32); t_ldm = time.time() - t0
        a = instantiate_from_config(ae_config(1, 128)); randomize_parameters(a, 1024, "fs.")
        t0 = time.time(); O.ae_decode(dict(a.state_dict()), torch.randn(1, 4, 64, 64)); t_dec = time.time() - t0
        c = instantiate_from_config(ae_config(2, 96)); randomize_parameters(c, 1024, "cs.")
        t0 = time.time(); O.ae_encode_mode(dict(c.state_dict()), torch.rand(1, 2, 512, 512)); t_enc = time.time() - t0
    per_volume = 250 * 8 * t_ccdm64 + 256 * (50 * t_ldm + t_dec + t_enc)
    return {"value": round(VOXELS_PER_VOLUME / per_volume, 1), "unit": "voxels/s", "cores": cores, "kind": "port",
            "sample": (f"oracle fp32: 1 CCDM UNet fwd @64^3 {t_ccdm64:.2f}s (x8 per 128^3 step, x250), 1 LDM UNet fwd N=1@64^2 {t_ldm:.2f}s (x50x256), "
                       f"1 AE decode {t_dec:.2f}s + 1 cond-encode {t_enc:.2f}s @512^2 (x256); linear extrapolation to one volume = {per_volume:.0f}s")}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs an MI355X (the product path has no CPU fallback)"
    if os.environ.get("GG_SINGLE_DEVICE") == "1":      # rehearsal of N > 1 ranks on a one-GPU box (all ranks share cuda:0)
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    from jointimagegeneration_amd import distributed as ggd
    ggd.init("nccl", device)
    from jointimagegeneration_amd import _lib
    from jointimagegeneration_amd.pipeline import GuideGenPipeline, build_ccdm, build_ldm
    _lib.load()
    torch.set_grad_enabled(False)

    def log(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    log("building models (random-init weights from the seed recipe)")
    pipe = GuideGenPipeline(build_ccdm(14, args.ccdm_steps, 1024, device), build_ldm(1024, device), ddim_steps=50)
    if os.environ.get("GG_NO_GRAPH") == "1":          # profiling aid: rocprofv3 --kernel-trace aborts on long hipGraph replays
        pipe.ccdm.use_graph = False
        pipe.sampler.use_graph = False
        pipe.use_graph = False
    log("models ready; untimed warm-up (weight repack, hipGraph capture)")

    def one_volume(i):
        return pipe.run_volume(N=args.volumes_per_gpu, mask_size=(128, 128, 128), depth=args.slices, hw=512, seed=1024 + rank + 1000 * i,
                               max_slices=args.max_slices)

    # untimed: weight repack + graph capture warm-up on the real shapes (2 short chains), then W full warm-up steps
    pipe.run_volume(N=args.volumes_per_gpu, mask_size=(128, 128, 128), depth=args.slices, hw=512, seed=7, ccdm_init_t=10005, max_slices=3)
    for i in range(args.warmup):
        one_volume(-1 - i)

    log(f"timed region: {args.steps} volume(s)")
    elapsed = ggd.timed_region(lambda: [one_volume(i) for i in range(args.steps)], device)

    if rank == 0:
        partial = args.max_slices is not None or args.ccdm_steps != 250 or args.slices != 256
        line = {
            "metric": "sampled voxels/sec @50 DDIM steps, 128^3 mask + 512^2x256 CT",
            "value": round(world * args.steps * args.volumes_per_gpu * VOXELS_PER_VOLUME / elapsed, 1), "unit": "voxels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 1),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "C5 full GuideGen volume per GPU: CCDM 128^3 K=14 250 steps -> LDM 256 slices x (cond-encode + 50 DDIM @4x64x64 + AE decode 512^2)",
                       "volumes_per_gpu_per_step": args.volumes_per_gpu, "ccdm_steps": args.ccdm_steps, "ddim_steps": 50, "slices": args.slices,
                       "parallelism": f"replicas x{world} (one volume per GPU, no collective)", "weights": "random-init (seed recipe)"},
        }
        if partial:
            line["partial"] = True
        line["stage_seconds"] = {k: round(v, 2) for k, v in pipe.stats.items()}
        log("measuring roofline / cpu baseline")
        if not args.no_roofline:
            line["roofline"] = conv_roofline(pipe, device)
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    ggd.finalize()


if __name__ == "__main__":
    main()
