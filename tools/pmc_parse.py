"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, as MI355X_MICROARCH.md prescribes) for the
3-D halo conv kernel:  python tools/pmc_parse.py <fetch_dir> <write_dir> <out.json>
Commands that produced the passes (GPU box):
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python tools/perf_probe.py ccdm128
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python tools/perf_probe.py ccdm128
gfx950 corrections: FETCH_SIZE counts 64 B per 128-B request for wide (16 B/lane) reads => doubled; WRITE_SIZE is exact
for 16-B stores; both are in KiB."""
import collections, csv, glob, hashlib, json, os, sys


def load(d, name):
    f = glob.glob(f"{d}/*/*_counter_collection.csv")[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name and "conv_halo_kernel<1" in r["Kernel_Name"]:
            k = (r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Grid_Size"])
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    return agg


F, W = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
n = sum(v[0] for v in F.values())
fetch_kb = sum(v[1] for v in F.values()) / n
write_kb = sum(v[1] for v in W.values()) / sum(v[0] for v in W.values())
out = {"kernel": "conv_halo_kernel<3-D> (all instantiations) in CCDM UNet forwards @128^3", "launches_profiled": n,
       "FETCH_SIZE_KiB_per_launch_raw": round(fetch_kb, 1), "WRITE_SIZE_KiB_per_launch": round(write_kb, 1),
       "traffic_bytes_per_launch": round((2 * fetch_kb + write_kb) * 1024),
       "correction": "FETCH_SIZE x2 (gfx950 wide-read under-count), WRITE_SIZE x1; separate --pmc passes",
       "per_shape": [{"kernel": k[0], "grid_threads": k[1], "launches": F[k][0], "fetch_KiB_raw": round(F[k][1] / F[k][0]),
                      "write_KiB": round(W[k][1] / W[k][0]) if k in W else None} for k in sorted(F, key=lambda k: -F[k][1])]}
# stamp WHICH kernel source the counters belong to: bench.py quotes this file as `roofline.traffic` and says whether the source it runs
# still is the one that was measured
src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "jointimagegeneration_amd", "csrc", "gg_conv_halo.hip")
out["kernel_source"] = "jointimagegeneration_amd/csrc/gg_conv_halo.hip"
out["kernel_source_sha256"] = hashlib.sha256(open(src, "rb").read()).hexdigest()
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out)[:600])
