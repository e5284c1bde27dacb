"""Per-kernel timeline of ONE eager LDM UNet forward from a rocprofv3 rocpd database:
   cd /tmp && GG_NO_GRAPH=1 rocprofv3 --kernel-trace --stats -d out -o ldm -- python3 tools/perf_probe.py ldm
   python tools/ldm_timeline.py out/ldm_results.db [--all]"""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name,start,end,grid_x,grid_y,grid_z,workgroup_x from kernels order by start"))
names = [r[0].split('(')[0] for r in rows]
n = len(rows)
per = next(p for p in range(150, 600) if names[n - p:] == names[n - 2 * p:n - p])
i = n - 6 * per                      # perf_probe: N=1 (1 warm + 5 timed) then N=4 (1 + 5)
while names[i - 1].startswith(('linear_f32', 'timestep', 'void at::')):
    i -= 1
f = rows[i - per:i]
print(f"{per} kernels per forward; sum of kernel durations {sum(e - s for _, s, e, *_ in f) / 1e3:.1f} us")
agg = collections.defaultdict(lambda: [0, 0])
for nm, s, e, *_ in f:
    k = nm.split('(')[0][:60]
    agg[k][0] += 1
    agg[k][1] += e - s
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:60s} {v[0]:4d} {v[1] / 1e3:8.1f} us  avg {v[1] / v[0] / 1e3:6.2f}")
if "--all" in sys.argv:
    t0 = f[0][1]
    for j, (nm, s, e, gx, gy, gz, wx) in enumerate(f):
        print(f"{j:3d} {(s - t0) / 1e3:8.1f} {(e - s) / 1e3:6.2f} {nm.split('(')[0][:44]:44s} grid {gx // wx}x{gy}x{gz}")
