"""Quick full-size timing probe (not a test): python tools/perf_probe.py [ccdm128|ccdm64|ldm|ae] ..."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if os.environ.get("H3_LIB"):      # same-box A/B: a variant library from tools/experiments/ab/
    from jointimagegeneration_amd import _lib
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "experiments", "ab", os.environ["H3_LIB"])
from jointimagegeneration_amd import ops
from jointimagegeneration_amd.ops import CL, pad32
from jointimagegeneration_amd.synth import randomize_parameters
from jointimagegeneration_amd.unet import UNetModel, create_unet_openai
from jointimagegeneration_amd.ldm import AutoencoderKL

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
which = sys.argv[1:] or ["ccdm64", "ldm", "ae"]

def timeit(fn, n=3, warm=1):
    for _ in range(warm): fn()
    best = 1e9
    for _ in range(int(os.environ.get("GG_PROBE_ROUNDS", "1"))):      # same-box A/B runs: best of several rounds
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(n): fn()
        torch.cuda.synchronize(); best = min(best, (time.time() - t0) / n)
    return best

def graph_time(fn) -> str:
    if not os.environ.get("GG_PROBE_GRAPH"):
        return ""
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return f", graph {timeit(lambda: g.replay(), n=5)*1e3:.3f} ms"


for w in which:
    if w.startswith("ccdm"):
        R = int(w[4:])
        u = create_unet_openai(image_size=128, in_channels=15, out_channels=14, num_res_blocks=2, cond_encoded_shape=None, dims=3,
                               base_channels=64, channel_mult=[1, 2, 2, 4, 5], attention_resolutions=[32, 16, 8], num_heads=1,
                               num_head_channels=32, softmax_output=True).eval()
        randomize_parameters(u, 1024, "ccdm."); u = u.to(dev)
        x = CL(torch.zeros(1, R, R, R, 32, dtype=torch.bfloat16, device=dev), 15)
        x.t[..., 0] = 1
        row = u.time_bias_rows(torch.tensor([17.0], device=dev))
        t = timeit(lambda: u.forward_cl(x, row), n=2)
        gf = 12720.7 * (R / 128) ** 3
        msg = f"CCDM UNet {R}^3 forward: {t*1e3:.2f} ms  ({gf/t/1e3:.1f} TFLOP/s)"
        if os.environ.get("GG_PROBE_GRAPH"):       # + the captured forward (what the sampler's captured step replays)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                u.forward_cl(x, row)
            tg = timeit(lambda: g.replay(), n=4)
            msg += f", graph {tg*1e3:.2f} ms"
        print(msg)
    elif w == "ldm":
        u = UNetModel(dims=2, image_size=512, in_channels=8, out_channels=4, model_channels=160, attention_resolutions=[8, 4, 2],
                      num_res_blocks=2, channel_mult=[1, 2, 4, 4, 5], num_head_channels=32).eval()
        randomize_parameters(u, 1024, "ldm."); u = u.to(dev)
        for N, R, gf in ((1, 64, 124.12), (4, 32, 118.48)):
            x = CL(torch.randn(N, 1, R, R, 32, device=dev).bfloat16(), 8)
            row = u.time_bias_rows(torch.full((N,), 981.0, device=dev))
            t = timeit(lambda: u.forward_cl(x, row), n=5)
            if os.environ.get("GG_NO_GRAPH"):
                print(f"LDM UNet N={N} {R}^2 forward: eager {t*1e3:.2f} ms")
                continue
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                u.forward_cl(x, row)
            tg = timeit(lambda: g.replay(), n=20)
            print(f"LDM UNet N={N} {R}^2 forward: eager {t*1e3:.2f} ms, graph {tg*1e3:.2f} ms ({gf/tg/1e3:.1f} TFLOP/s)")
    elif w == "ae":
        for name, inch, ch, gfd, gfe in (("first_stage", 1, 128, 2513.3, None), ("cond_stage", 2, 96, None, 634.5)):
            a = AutoencoderKL(ddconfig=dict(double_z=True, z_channels=4, resolution=512, in_channels=inch, out_ch=inch, ch=ch,
                                            ch_mult=[1, 2, 4, 4], num_res_blocks=2, dropout=0.0, dims=2, attn_resolutions=[16, 8]),
                              embed_dim=4, dims=2).eval()
            randomize_parameters(a, 1024, name + "."); a = a.to(dev)
            if gfd:
                z = CL(torch.randn(1, 1, 64, 64, 32, device=dev).bfloat16(), 4)
                t = timeit(lambda: a.decode_cl(z), n=3)
                print(f"AE decode 64^2->512^2: {t*1e3:.3f} ms ({gfd/t/1e3:.1f} TFLOP/s)" + graph_time(lambda: a.decode_cl(z)))
            if gfe:
                x = CL(torch.randn(1, 1, 512, 512, 32, device=dev).bfloat16(), 2)
                t = timeit(lambda: a.encode_moments_cl(x), n=3)
                gt = graph_time(lambda: a.encode_moments_cl(x))
                print(f"AE cond-encode 512^2->64^2: {t*1e3:.3f} ms ({gfe/t/1e3:.1f} TFLOP/s)" + gt)
