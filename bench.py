#!/usr/bin/env python3
"""GuideGen sampling benchmark on MI355X (contract in the task statement; metric from BASELINE.json).

  python bench.py --gpus N --steps K --warmup W

A "step" is one complete GuideGen volume per GPU (config C5 of BASELINE.json, one independent volume per rank):
  CCDM 3-D categorical UNet, 128^3 mask, 14 classes, 250 reverse steps  ->  stage glue  ->
  LDM autoregressive CT, 256 slices of 512x512: per slice cond-stage encode + 50 DDIM steps (latent 4x64x64) + AE decode.
value = sampled voxels/s = (N * steps * 69 206 016 voxels) / wall time of the timed volumes (max over ranks), synthetic inputs
and random-init weights of the reference architectures (no checkpoints or data exist offline), inputs resident in HBM.
Volumes are independent => weak scaling, no collective on the data path.

Wall-clock budget.  One volume takes ~20-30 s, so `--steps 20 --warmup 5` would not fit the driver's 600 s limit.  --steps and
--warmup are therefore UPPER BOUNDS under a budget (`--budget-s`, default 420 s counted from process start): the untimed
warm-up is the short capture run (weight repack + hipGraph capture on the real shapes) plus as many full warm-up volumes as
requested AND affordable (at most one), and the timed region runs whole volumes - never truncated slices or steps - until
`--steps` is reached or the next volume would not fit.  The line reports `steps` = volumes actually timed, `steps_requested`,
`warmup` = full warm-up volumes actually run, `warmup_requested`.

N > 1.  Under a launcher (torch.distributed.run: RANK/LOCAL_RANK/WORLD_SIZE set) each process is one rank.  Started directly
(`python bench.py --gpus 4`, WORLD_SIZE unset) the process becomes a PARENT that never touches the GPU: it starts N child
ranks (one per LOCAL_RANK, MASTER_ADDR=127.0.0.1), forwards rank 0's JSON line and exits non-zero if any child fails.
"""
from __future__ import annotations

import os
import sys
import time

T_START = float(os.environ.get("GG_BENCH_T0", "0")) or time.time()

import argparse  # noqa: E402
import json  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VOXELS_PER_VOLUME = 128 ** 3 + 512 * 512 * 256          # 69 206 016
MFMA_PEAK_TFLOPS = 2500.0                                # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0
CCDM_UNET_GFLOP_128 = 12720.7                            # SURVEY.md 8d (torch flop counter over the reference modules)
LDM_UNET_GFLOP = 124.12                                  # N=1 @64x64
LDM_UNET_WEIGHT_MB = 535.0                               # 267.5 M params in bf16: streamed once per forward
AE_DECODE_GFLOP = 2513.31
COND_ENCODE_GFLOP = 634.51


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4, help="upper bound on timed volumes per GPU (see --budget-s)")
    ap.add_argument("--warmup", type=int, default=0, help="upper bound on full warm-up volumes (at most one is run)")
    ap.add_argument("--budget-s", type=float, default=float(os.environ.get("GG_BENCH_BUDGET_S", "420")),
                    help="wall-clock budget of the whole process, counted from its start")
    ap.add_argument("--ccdm-steps", type=int, default=250, help="CCDM reverse steps (250 = params_eval.yml time_steps)")
    ap.add_argument("--slices", type=int, default=256)
    ap.add_argument("--volumes-per-gpu", type=int, default=1, help="independent volumes sampled concurrently per GPU (BASELINE C5 = 1)")
    ap.add_argument("--max-slices", type=int, default=None, help="DEV ONLY: truncate the slice loop (marks the line partial)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary 8-volumes-per-GPU sample")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ parent launcher
def launch_ranks(args) -> int:
    """Parent of a direct `--gpus N` run: no torch import, no GPU call.  One child per rank; rank 0's stdout is ours."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GG_BENCH_T0=repr(T_START), GG_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out = None if r == 0 else subprocess.DEVNULL          # ranks > 0 print nothing on stdout; stderr is shared
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out))
    rc = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in pending:                             # a failed rank would leave the others in a barrier forever
                    q.terminate()
        time.sleep(0.2)
    return rc


# ------------------------------------------------------------------------------------------------ roofline legs
def measured_peaks(device):
    """What THIS box delivers (SURVEY.md 8d last sentence): a register-resident bf16 MFMA loop on every CU (random
    non-zero operands, ~0.3 s per shape after a warm-up launch, so the chip is at the clock it holds under matrix load) and a
    16-bytes-per-lane stream copy of 1 GiB (read + write bytes).  HIP events on the launching stream; untimed part of the bench."""
    import ctypes
    import torch
    from jointimagegeneration_amd import _lib
    lib = _lib.load()
    stream = torch.cuda.current_stream().cuda_stream
    sink = torch.zeros(4, device=device)
    out = {}

    def mfma(shape, wps, iters):
        fl = ctypes.c_double(0.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(lib.gg_ubench_mfma_bf16(shape, iters, wps, sink.data_ptr(), ctypes.byref(fl), stream), "gg_ubench_mfma_bf16")
        e1.record()
        torch.cuda.synchronize()
        return fl.value / (e0.elapsed_time(e1) * 1e-3) / 1e12, e0.elapsed_time(e1)

    best = {}
    for shape, name in ((0, "16x16x32"), (1, "32x32x16")):
        for wps in (1, 2):
            mfma(shape, wps, 20000)                                       # warm-up (code load, clock ramp)
            tf, ms = mfma(shape, wps, 600000 // wps)                      # ~150-300 ms of back-to-back MFMAs per SIMD
            if tf > best.get(name, (0, 0, 0))[0]:
                best[name] = (tf, wps, ms)
    out["mfma_tflops"] = round(max(v[0] for v in best.values()), 1)
    out["mfma_by_shape"] = {k: {"tflops": round(v[0], 1), "waves_per_simd": v[1], "ms": round(v[2], 1)} for k, v in best.items()}
    nbytes = 1 << 30
    src = torch.empty(nbytes, dtype=torch.uint8, device=device).random_(0, 255)
    dst = torch.empty_like(src)
    rates = []
    for i in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(lib.gg_ubench_stream_copy(src.data_ptr(), dst.data_ptr(), nbytes, stream), "gg_ubench_stream_copy")
        e1.record()
        torch.cuda.synchronize()
        if i:
            rates.append(2.0 * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    out["hbm_gbs"] = round(max(rates), 1)
    out["how"] = ("mfma: register-resident v_mfma bf16 loop on every SIMD (gg_ubench_mfma_bf16, random operands, best of 1 / 2 waves per SIMD "
                  "and of the two shapes); hbm: 1 GiB float4 grid-stride copy (gg_ubench_stream_copy), read + write bytes, best of 5")
    del src, dst
    torch.cuda.empty_cache()
    return out


def add_measured_fractions(roofline, peaks):
    """frac_vs_measured beside every frac: the same achieved figure over the box's own MFMA / copy rate instead of the vendor peak."""
    def one(e):
        if not isinstance(e, dict) or "achieved" not in e or "unit" not in e:
            return
        peak = peaks["mfma_tflops"] if e["unit"] == "TFLOP/s" else peaks["hbm_gbs"]
        e["frac_vs_measured"] = round(e["achieved"] / peak, 4)
    one(roofline)
    for e in roofline.get("stages", {}).values():
        one(e)
    if isinstance(roofline.get("all_3x3x3_convs"), dict) and "achieved" in roofline["all_3x3x3_convs"]:
        roofline["all_3x3x3_convs"]["frac_vs_measured"] = round(roofline["all_3x3x3_convs"]["achieved"] / peaks["mfma_tflops"], 4)
    roofline["measured_peak"] = peaks


def conv_roofline(pipe, device):
    """Judged kernel = the 3-D halo-tile implicit-GEMM conv (`conv_halo_kernel<1,NT,UP>`) of the CCDM UNet
    (3x3x3 convs are 99.3 % of the UNet's FLOPs; the halo kernel runs all of them at the 128^3..32^3 levels).
    One eager 128^3 UNet forward with a HIP event pair around EVERY conv launch, recorded on the stream the kernels are
    launched on; achieved = sum(algorithmic FLOPs of the halo launches) / sum(their durations).  The rocprofv3 average of
    the same kernel name (profiles/) must agree with avg_launch_ms.  traffic: PMC passes parsed by tools/pmc_parse.py."""
    import torch
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
        halo = ops.conv_runs_halo_tile(src1, cout, k=k, stride=stride, pad=pad, upsample=upsample, src2=src2)
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
        ef0, ef1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ef0.record()
        unet.forward_cl(x, row)
        ef1.record()
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
           "all_3x3x3_convs": {"launches": len(k27), "achieved": round(fl27 / t27 / 1e12, 1), "flops_per_forward": fl27},
           "eager_forward_ms_with_event_pairs": round(ef0.elapsed_time(ef1), 2)}
    for name in ("r04/pmc_conv3d.json", "r03/final_pmc_conv3d.json", "r03/pmc_conv3d.json", "r02/pmc_conv3d.json", "r01_pmc_conv3d.json"):
        pmc = os.path.join(ROOT, "profiles", name)
        if os.path.exists(pmc):
            j = json.load(open(pmc))
            out["traffic"] = j["traffic_bytes_per_launch"]
            out["traffic_source"] = f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2 gfx950 correction)"
            # the counters belong to ONE version of the kernel: say whether the source this run was built from still is that version
            import hashlib
            src = os.path.join(ROOT, "jointimagegeneration_amd", "csrc", "gg_conv_halo.hip")
            sha = hashlib.sha256(open(src, "rb").read()).hexdigest() if os.path.exists(src) else None
            out["traffic_kernel_source_sha256"] = j.get("kernel_source_sha256")
            out["traffic_matches_this_source"] = (sha is not None and sha == j.get("kernel_source_sha256"))
            break
    return out


def stage_rooflines(pipe, ccdm_ms_per_step):
    """The stages that make up the wall time of a volume, each against ITS roofline t_roof = max(FLOP / MFMA peak, bytes / HBM
    peak): HIP events around the captured hipGraph replays of one slice (cond-encode, 50 DDIM steps, decode)."""
    tm = pipe.time_slice_stages()
    out = {}

    def entry(name, ms, gflop, mbytes, what):
        t_mfma, t_hbm = gflop / (MFMA_PEAK_TFLOPS * 1e3) * 1e3, mbytes / (HBM_PEAK_GBS * 1e3) * 1e3     # ms
        bound = "mfma" if t_mfma >= t_hbm else "hbm"
        e = {"bound": bound, "what": what, "ms": round(ms, 4), "t_roof_ms": round(max(t_mfma, t_hbm), 4),
             "frac": round(max(t_mfma, t_hbm) / ms, 4), "mfma_frac": round(t_mfma / ms, 4)}
        if bound == "hbm":
            e.update(achieved=round(mbytes / ms, 1), peak=HBM_PEAK_GBS, unit="GB/s")
        else:
            e.update(achieved=round(gflop / ms, 1), peak=MFMA_PEAK_TFLOPS, unit="TFLOP/s")
        out[name] = e

    entry("ldm_unet_step_n1_64x64", tm["ddim_step_ms"], LDM_UNET_GFLOP, LDM_UNET_WEIGHT_MB,
          "one captured DDIM step = latent UNet forward (N=1, 8x64x64 in) + fused update; bytes = bf16 weights streamed once")
    entry("ae_decode_512", tm["decode_ms"], AE_DECODE_GFLOP, 99.0 + 2 * 67.0, "AutoencoderKL.decode 4x64x64 -> 512x512 (captured graph)")
    entry("cond_encode_512", tm["encode_ms"], COND_ENCODE_GFLOP, 38.0 + 2 * 50.0, "cond-stage AutoencoderKL.encode 2x512x512 -> 8x64x64 (captured graph)")
    if ccdm_ms_per_step:
        entry("ccdm_unet_step_128", ccdm_ms_per_step, CCDM_UNET_GFLOP_128, 190.8 + 15000.0,
              "one captured CCDM reverse step @128^3 (UNet forward + fused posterior/sample), from the timed volumes")
    return out


def other_config_stages(device):
    """roofline.stages entries of the two shipped / named configurations that are NOT on the C5 wall, each one
    captured forward at batch 1 timed with HIP events over 20 graph replays:
      * the latent UNet with SpatialTransformer cross-attention (BASELINE.json C4 as worded: context 512 x 768, 365.2 M params,
        181.5 GFLOP per forward; HBM-bound: 730 MB of bf16 weights streamed once per forward),
      * the pixel-space UNet of configs/latent-diffusion/ruijin-ldm_from_controlnet.yaml (3 x 512^2 in, 172.9 M params, 4 629 GFLOP per
        forward; MFMA-bound)."""
    import torch
    from jointimagegeneration_amd import ops
    from jointimagegeneration_amd.ops import CL
    from jointimagegeneration_amd.pipeline import LDM_UNET
    from jointimagegeneration_amd.synth import randomize_parameters
    from jointimagegeneration_amd.unet import UNetModel
    out = {}

    def timed(unet, x, row, ctx):
        unet.forward_cl(x, row, ctx)
        torch.cuda.synchronize()
        g = ops.capture_graph(lambda: unet.forward_cl(x, row, ctx))
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 20

    def entry(name, ms, gflop, mbytes, what):
        t_mfma, t_hbm = gflop / (MFMA_PEAK_TFLOPS * 1e3) * 1e3, mbytes / (HBM_PEAK_GBS * 1e3) * 1e3
        bound = "mfma" if t_mfma >= t_hbm else "hbm"
        e = {"bound": bound, "what": what, "ms": round(ms, 4), "t_roof_ms": round(max(t_mfma, t_hbm), 4), "frac": round(max(t_mfma, t_hbm) / ms, 4),
             "mfma_frac": round(t_mfma / ms, 4), "on_c5_wall": False}
        e.update(dict(achieved=round(mbytes / ms, 1), peak=HBM_PEAK_GBS, unit="GB/s") if bound == "hbm" else
                 dict(achieved=round(gflop / ms, 1), peak=MFMA_PEAK_TFLOPS, unit="TFLOP/s"))
        out[name] = e

    u = UNetModel(**dict(LDM_UNET["params"], use_spatial_transformer=True, context_dim=768, transformer_depth=1)).eval()
    randomize_parameters(u, 1024, "ldm_st.")
    u = u.to(device)
    x = CL(torch.randn(1, 1, 64, 64, 32, device=device).bfloat16(), 8)
    ctx = u.context_cl(torch.randn(1, 512, 768, device=device))
    ms = timed(u, x, u.time_bias_rows(torch.full((1,), 481.0, device=device)), ctx)
    entry("ldm_st_unet_forward_n1_64x64_ctx512x768", ms, 181.52, 730.4, "latent UNet + SpatialTransformer cross-attention (C4 as named), one captured forward")
    del u, x, ctx
    u = UNetModel(dims=2, image_size=512, in_channels=3, out_channels=1, model_channels=128, attention_resolutions=[32, 16, 8], num_res_blocks=2,
                  channel_mult=[1, 2, 4, 4, 5], num_head_channels=32).eval()
    randomize_parameters(u, 1024, "ldm_pixel.")
    u = u.to(device)
    x = CL(torch.randn(1, 1, 512, 512, 32, device=device).bfloat16(), 3)
    ms = timed(u, x, u.time_bias_rows(torch.full((1,), 481.0, device=device)), None)
    entry("pixel_unet_forward_n1_3x512x512", ms, 4629.0, 345.8 + 2 * 1200.0, "pixel-space UNet of ruijin-ldm_from_controlnet.yaml, one captured forward")
    del u, x
    torch.cuda.empty_cache()
    return out


def wall_shares(stages, volume_ms, ccdm_steps, slices, ddim_steps=50):
    """Share of one volume's wall time spent in each stage (launch count x captured stage time / measured volume time), and the
    stage that dominates the wall with ITS roofline fraction: the top-level `roofline` names the MFMA-bound judged kernel, which is
    a minority of the wall; this is the entry that says where the time goes."""
    count = {"ldm_unet_step_n1_64x64": slices * ddim_steps, "ae_decode_512": slices, "cond_encode_512": slices, "ccdm_unet_step_128": ccdm_steps}
    shares = {k: round(count[k] * v["ms"] / volume_ms, 4) for k, v in stages.items() if k in count}
    top = max(shares, key=shares.get)
    return shares, {"stage": top, "wall_share": shares[top], "bound": stages[top]["bound"], "frac": stages[top]["frac"],
                    "ms": stages[top]["ms"], "t_roof_ms": stages[top]["t_roof_ms"], "launches_per_volume": count[top]}


def cpu_baseline():
    """Oracle (CPU fp32 restatement, validated against the reference in the build container) timed on the host cores on a
    bounded sample of the same workload and extrapolated linearly: one CCDM UNet forward at 64^3 (x8 -> 128^3), one LDM
    UNet forward (N=1, 64^2), one AE decode and one cond-stage encode at 512^2."""
    import torch
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


def batch8_sample(pipe, args):
    """NOT the headline (BASELINE C5 is one volume per GPU): what the same GPU delivers when 8 independent volumes are sampled as ONE
    batch (N = 8 through every kernel).  Bounded sample, extrapolated linearly and marked so: captured CCDM steps from the difference
    of a 20- and a 10-step chain, captured slices from the difference of an 8- and a 4-slice loop (capture / eager warm-up cancel)."""
    import torch
    N, size = 8, (128, 128, 128)

    def ccdm(steps):
        torch.cuda.synchronize(); t0 = time.time()
        lab = pipe.sample_mask(N, size, 12, init_t=10000 + steps)
        torch.cuda.synchronize()
        return time.time() - t0, lab

    def ldm(lab, slices):
        torch.cuda.synchronize(); t0 = time.time()
        pipe.sample_ct(lab, args.slices, 512, 13, max_slices=slices)
        torch.cuda.synchronize()
        return time.time() - t0

    pipe.run_volume(N=N, mask_size=size, depth=args.slices, hw=512, seed=11, ccdm_init_t=10005, max_slices=3)     # repack / capture at N = 8
    t10, _ = ccdm(10)
    t20, lab = ccdm(20)
    t10, t20 = min(t10, ccdm(10)[0]), min(t20, ccdm(20)[0])          # (the first pair carries allocator / capture noise: 115-168 ms per step run to run)
    s4 = ldm(lab, 4)
    s8 = ldm(lab, 8)
    step_s, slice_s = (t20 - t10) / 10.0, (s8 - s4) / 4.0
    per_batch = args.ccdm_steps * step_s + args.slices * slice_s
    return {"value": round(N * VOXELS_PER_VOLUME / per_batch, 1), "unit": "voxels/s", "headline": False, "extrapolated": True,
            "ccdm_step_ms_n8": round(step_s * 1e3, 2), "ldm_slice_ms_n8": round(slice_s * 1e3, 2), "seconds_per_8_volumes": round(per_batch, 1),
            "sample": "N=8 batch: (20-step - 10-step CCDM chain, best of two each)/10, (8-slice - 4-slice loop)/4, x250 steps + x256 slices"}


# ------------------------------------------------------------------------------------------------ rank body
def agree(flag: bool) -> bool:
    """All ranks take the same go / stop decision (logical AND); a 1-element host-side reduce, off the data path."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return flag
    on_gpu = dist.get_backend() == "nccl"
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device="cuda" if on_gpu else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def agree_min(v: int) -> int:
    """The smallest of the ranks' values (every rank must time the same number of volumes); host-side, outside the timed region."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return v
    on_gpu = dist.get_backend() == "nccl"
    t = torch.tensor([v], dtype=torch.int32, device="cuda" if on_gpu else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item())


def main():
    args = parse()
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if world_env is None and args.gpus > 1:
        sys.exit(launch_ranks(args))                       # parent: spawns the ranks, never touches the GPU
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher must start exactly --gpus ranks")

    import torch
    dry = os.environ.get("GG_BENCH_DRY") == "1"           # launcher / protocol rehearsal on CPU ranks (tests): no sampling at all
    from jointimagegeneration_amd import distributed as ggd

    def log(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')} +{time.time() - T_START:5.0f}s] {msg}", file=sys.stderr, flush=True)

    if dry:
        device = None
        ggd.init("gloo")
        pipe = None
    else:
        assert torch.cuda.is_available(), "bench.py needs an MI355X (the product path has no CPU fallback)"
        if os.environ.get("GG_SINGLE_DEVICE") == "1":      # rehearsal of N > 1 ranks on a one-GPU box (all ranks share cuda:0)
            local = 0
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
        ggd.init("nccl", device)
        from jointimagegeneration_amd import _lib
        from jointimagegeneration_amd.pipeline import GuideGenPipeline, build_ccdm, build_ldm
        _lib.load()
        torch.set_grad_enabled(False)
        log("building models (random-init weights from the seed recipe)")
        pipe = GuideGenPipeline(build_ccdm(14, args.ccdm_steps, 1024, device), build_ldm(1024, device), ddim_steps=50)
        if os.environ.get("GG_NO_GRAPH") == "1":          # profiling aid (rocprofv3 --kernel-trace on long hipGraph replays)
            pipe.ccdm.use_graph = False
            pipe.sampler.use_graph = False
            pipe.use_graph = False

    def one_volume(i):
        if dry:
            time.sleep(0.05)
            return
        pipe.run_volume(N=args.volumes_per_gpu, mask_size=(128, 128, 128), depth=args.slices, hw=512, seed=1024 + rank + 1000 * i,
                        max_slices=args.max_slices)

    # ---- untimed: weight repack + hipGraph capture on the real shapes (a short chain and 3 slices)
    if not dry:
        log("untimed warm-up (weight repack, hipGraph capture)")
        pipe.run_volume(N=args.volumes_per_gpu, mask_size=(128, 128, 128), depth=args.slices, hw=512, seed=7, ccdm_init_t=10005, max_slices=3)

    # ---- roofline + CPU baseline legs BEFORE the timed region, so that what is left of the budget is known exactly
    line_extra = {}
    if rank == 0 and not dry:
        if not args.no_roofline:
            log("roofline legs (HIP events)")
            line_extra["roofline"] = conv_roofline(pipe, device)
            log("measured peaks (MFMA microbenchmark, stream copy)")
            try:
                line_extra["measured_peak"] = measured_peaks(device)
            except Exception as e:                                           # never lose the line to an auxiliary leg
                line_extra["measured_peak"] = {"error": repr(e)}
        # single-GPU characterisation legs (~50 s): only at N = 1, so that in a multi-rank run no rank waits for rank 0 in a barrier
        if not args.no_roofline and not args.no_extra and world == 1:
            log("roofline legs of the configurations off the C5 wall (SpatialTransformer UNet, pixel-space UNet)")
            try:
                line_extra["other_stages"] = other_config_stages(device)
            except Exception as e:                                           # never lose the line to an auxiliary leg
                line_extra["other_stages"] = {"error": repr(e)}
        if args.volumes_per_gpu == 1 and not args.no_extra and args.max_slices is None and world == 1:
            log("secondary leg: 8 volumes per GPU (bounded sample)")
            try:
                line_extra["extra"] = {"volumes_per_gpu_8": batch8_sample(pipe, args)}
            except Exception as e:                                           # never lose the line to an auxiliary leg
                line_extra["extra"] = {"volumes_per_gpu_8": {"error": repr(e)}}

    # after the timed region: stage timings + JSON + teardown, and on rank 0 the CPU-baseline leg (~35 s of host work: it runs AFTER the
    # timed region and after the last collective, so that the other ranks never wait in a barrier for it)
    TAIL_S = 12.0 + (0.0 if (args.no_cpu_baseline or dry) else 25.0)
    est = None                                               # seconds per volume, measured
    warm_done = 0
    ggd.barrier(device)
    if args.warmup > 0:
        # a full warm-up volume is affordable only if at least two more volumes fit after it (est. from the previous round: 30 s)
        guess = 0.05 if dry else float(os.environ.get("GG_BENCH_VOLUME_GUESS_S", "30"))
        if agree(time.time() - T_START + 3 * guess * 1.1 + TAIL_S <= args.budget_s):
            t0 = time.time()
            one_volume(-1)
            if device is not None:
                torch.cuda.synchronize()
            est = time.time() - t0
            warm_done = 1

    # ---- how many whole volumes fit: decided ONCE, before the timed region, from the measured warm-up volume (or the guess), and agreed
    #      over the ranks by one MIN-reduce, so that the timed region itself holds no collective at all
    per_guess = est if est is not None else (0.05 if dry else float(os.environ.get("GG_BENCH_VOLUME_GUESS_S", "26")))
    left = args.budget_s - (time.time() - T_START) - TAIL_S
    planned = max(1, min(args.steps, int(left / (per_guess * 1.04))))
    planned = agree_min(planned)
    log(f"timed region: {planned} volume(s) (requested {args.steps}; {per_guess:.1f} s per volume, {args.budget_s:.0f} s budget)")
    done = 0
    ggd.barrier(device)
    if device is not None:
        torch.cuda.synchronize()
    t_region = time.time()
    while done < planned:
        one_volume(done)
        done += 1
    if device is not None:
        torch.cuda.synchronize()
    ggd.barrier(device)
    elapsed = time.time() - t_region
    if world > 1:
        import torch.distributed as dist
        on_gpu = dist.get_backend() == "nccl"
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        partial = args.max_slices is not None or args.ccdm_steps != 250 or args.slices != 256
        line = {
            "metric": "sampled voxels/sec @50 DDIM steps, 128^3 mask + 512^2x256 CT",
            "value": round(world * done * args.volumes_per_gpu * VOXELS_PER_VOLUME / elapsed, 1), "unit": "voxels/s",
            "n_gpus": world, "steps": done, "warmup": warm_done, "ms_per_step": round(elapsed / done * 1e3, 1),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "steps_requested": args.steps, "warmup_requested": args.warmup, "budget_s": args.budget_s,
            "warmup_note": "untimed capture run (5 CCDM steps + 3 slices on the real shapes) + `warmup` full volumes; --steps/--warmup are upper bounds under budget_s",
            "config": {"workload": "C5 full GuideGen volume per GPU: CCDM 128^3 K=14 250 steps -> LDM 256 slices x (cond-encode + 50 DDIM @4x64x64 + AE decode 512^2)",
                       "volumes_per_gpu_per_step": args.volumes_per_gpu, "ccdm_steps": args.ccdm_steps, "ddim_steps": 50, "slices": args.slices,
                       "parallelism": f"replicas x{world} (one volume per GPU, no collective)", "weights": "random-init (seed recipe)"},
        }
        if partial:
            line["partial"] = True
        if dry:
            line["dry_run"] = True
            line["value"] = 0.0
        else:
            line["stage_seconds"] = {k: round(v, 2) for k, v in pipe.stats.items()}
            if not args.no_cpu_baseline:
                log("CPU baseline leg (oracle on the host cores, bounded sample)")
                line_extra["cpu_baseline"] = cpu_baseline()
            line.update(line_extra)
            if "roofline" in line and not partial:
                try:
                    st = stage_rooflines(pipe, pipe.stats["ccdm_s"] / args.ccdm_steps * 1e3)
                    st.update(line.pop("other_stages", {}))
                    line["roofline"]["stages"] = st
                    line["roofline"]["wall_shares"], line["roofline"]["dominant_by_wall"] = wall_shares(st, elapsed / done * 1e3, args.ccdm_steps, args.slices)
                except Exception as e:                                       # never lose the line to an auxiliary leg
                    line["roofline"]["stages_error"] = repr(e)
            peaks = line.pop("measured_peak", None)
            if "roofline" in line and peaks is not None:
                if "error" in peaks:
                    line["roofline"]["measured_peak"] = peaks
                else:
                    add_measured_fractions(line["roofline"], peaks)
        line["wall_s_total"] = round(time.time() - T_START, 1)
        print(json.dumps(line), flush=True)
    ggd.finalize()


if __name__ == "__main__":
    main()
