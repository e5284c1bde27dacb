"""Copy the summaries of tools/profile_round.sh (gpurun_out/<dir>) into profiles/<round>/ and write the judged-kernel cross-check:
python tools/collect_profiles.py gpurun_out/prof_r04 profiles/r04 profiles/r04/bench_driver_cmd_v3.json"""
import csv, json, os, shutil, sys
src, dst, bench_json = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(dst, exist_ok=True)
for a, b in (("pmc_conv3d.json", "pmc_conv3d.json"), ("bench/bench_kernel_stats.csv", "bench_kernel_stats.csv"), ("ccdm/ccdm_kernel_stats.csv", "ccdm_kernel_stats.csv"),
             ("ae/ae_kernel_stats.csv", "ae_kernel_stats.csv"), ("pmc_sq_summary.txt", "pmc_sq_summary.txt"), ("pmc_ae_summary.txt", "pmc_ae_summary.txt"),
             ("ldm_unet_forward_timeline_eager.txt", "ldm_unet_forward_timeline_eager.txt")):
    shutil.copy(os.path.join(src, a), os.path.join(dst, b))
rows = list(csv.DictReader(open(os.path.join(dst, "bench_kernel_stats.csv"))))
h = [r for r in rows if r["Name"].startswith("void conv_halo_kernel<1,")]
calls = sum(int(r["Calls"]) for r in h)
tot = sum(float(r["TotalDurationNs"]) for r in h)
post = [r for r in h if r["Name"].startswith("void conv_halo_kernel<1, 1, 0, 2, 1>")]
plain = [r for r in h if r["Name"].startswith("void conv_halo_kernel<1, 1, 0, 2, 0>")]
d = json.loads(open(bench_json).read().strip().splitlines()[-1])["roofline"]
out = ["rocprofv3 --kernel-trace --stats of `python3 bench.py --gpus 1 --steps 1 --warmup 0 --ccdm-steps 20 --max-slices 6 --no-cpu-baseline --no-extra` (tools/profile_round.sh pass 1)"]
for r in h:
    out.append(f"  {r['Name'][:60]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs']) / 1e3:8.1f} us")
avg = tot / calls / 1e3
out.append(f"3-D halo conv, all instantiations: {calls} launches, avg {avg:.1f} us = {345.614 / avg * 1e3:.1f} TF/s = {345.614 / avg * 1e3 / 2500:.4f} of 2.5 PF at 345.614 GFLOP per launch")
out.append(f"bench line (HIP events, the driver's command, {os.path.basename(bench_json)}): avg_launch_ms {d['avg_launch_ms']} -> {d['achieved']} TF/s = frac {d['frac']}, frac_vs_measured {d['frac_vs_measured']}")
if post and plain:
    np_, dp, dq = int(post[0]["Calls"]), float(post[0]["AverageNs"]) / 1e3, float(plain[0]["AverageNs"]) / 1e3
    out.append(f"Note: {np_} of the {calls} launches are the head conv WITH the CCDM reverse step as its epilogue (<1, 1, 0, 2, 1>: {dp:.1f} us, of which the conv is the")
    out.append(f"{dq:.1f} us of <1, 1, 0, 2, 0>); the bench line's 35 launches are one eager UNet forward, whose head conv carries no reverse step.  Counting those")
    out.append(f"{np_} launches at the plain head conv's duration: avg {(calls * avg - np_ * (dp - dq)) / calls:.1f} us.")
open(os.path.join(dst, "judged_kernel_check.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
