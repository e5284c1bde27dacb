# instructions per wave of the CCDM forward's kernels (rocprofv3 PMC pass)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_insts_ccdm
rm -rf $OUT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA --kernel-trace --output-format csv -d $OUT -o p -- python3 $GRAFT_REPO_ROOT/tools/perf_probe.py ccdm128 > $OUT.log 2>&1
python3 - <<'PY'
import csv, glob, os, re, collections
fs = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_insts_ccdm/**/*counter_collection.csv", recursive=True)
if not fs: print(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_insts_ccdm.log").read()[-2000:]); raise SystemExit
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(fs[0])):
    k = re.sub(r"\(.*", "", r["Kernel_Name"])[:44] + " g" + r["Grid_Size"]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
print(f"{'kernel':56s} {'n':>4s} {'waves':>8s} {'VALU/w':>8s} {'MFMA/w':>8s} {'SALU/w':>8s} {'LDS/w':>7s} {'VMEM/w':>7s}")
rows = []
for k, c in acc.items():
    w = c["SQ_WAVES"] or 1
    rows.append((c["SQ_INSTS_VALU"] + c["SQ_INSTS_SALU"], k, cnt[k], w / cnt[k], c["SQ_INSTS_VALU"] / w, c["SQ_INSTS_MFMA"] / w, c["SQ_INSTS_SALU"] / w, c["SQ_INSTS_LDS"] / w, c["SQ_INSTS_VMEM_RD"] / w))
for r in sorted(rows, reverse=True)[:14]:
    print(f"{r[1]:56s} {r[2]:4d} {r[3]:8.0f} {r[4]:8.0f} {r[5]:8.0f} {r[6]:8.0f} {r[7]:7.0f} {r[8]:7.0f}")
PY
