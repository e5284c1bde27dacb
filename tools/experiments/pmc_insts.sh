# instructions per wave of every kernel of the eager latent-UNet forward (rocprofv3 PMC pass; separate from timing runs)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_insts
rm -rf $OUT
GG_NO_GRAPH=1 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT -o p -- python3 $GRAFT_REPO_ROOT/tools/perf_probe.py ldm > $OUT.log 2>&1
python3 - <<'PY'
import csv, glob, os, re, collections
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_insts/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = re.sub(r"\(.*", "", r["Kernel_Name"])[:50] + " g" + r["Grid_Size"]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
print(f"{'kernel':60s} {'n':>4s} {'waves':>7s} {'VALU/w':>8s} {'SALU/w':>8s} {'SMEM/w':>7s} {'LDS/w':>7s} {'VMEM/w':>7s} {'total/w':>8s}")
rows = []
for k, c in acc.items():
    w = c["SQ_WAVES"] or 1
    tot = (c["SQ_INSTS_VALU"] + c["SQ_INSTS_SALU"] + c["SQ_INSTS_SMEM"] + c["SQ_INSTS_LDS"] + c["SQ_INSTS_VMEM_RD"] + c["SQ_INSTS_VMEM_WR"]) / w
    rows.append((cnt[k] * tot, k, cnt[k], w / cnt[k], c["SQ_INSTS_VALU"] / w, c["SQ_INSTS_SALU"] / w, c["SQ_INSTS_SMEM"] / w, c["SQ_INSTS_LDS"] / w, (c["SQ_INSTS_VMEM_RD"] + c["SQ_INSTS_VMEM_WR"]) / w, tot))
for r in sorted(rows, reverse=True)[:45]:
    print(f"{r[1]:60s} {r[2]:4d} {r[3]:7.0f} {r[4]:8.0f} {r[5]:8.0f} {r[6]:7.1f} {r[7]:7.0f} {r[8]:7.0f} {r[9]:8.0f}")
PY
