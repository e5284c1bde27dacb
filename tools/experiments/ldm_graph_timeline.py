"""Timeline of the LAST graph replay in a rocprofv3 database of probe_ldm_graph.py: per kernel its duration and the gap to its
predecessor's end.   python tools/experiments/ldm_graph_timeline.py out/ldmg_results.db [--all]"""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name,start,end,grid_x,grid_y,grid_z,workgroup_x from kernels order by start"))
names = [r[0].split('(')[0] for r in rows]
n = len(rows)
per = next(p for p in range(100, 600) if names[n - p:] == names[n - 2 * p:n - p])
f = rows[n - per:]
dur = sum(e - s for _, s, e, *_ in f) / 1e3
span = (f[-1][2] - f[0][1]) / 1e3
gaps = [(f[j][1] - f[j - 1][2]) / 1e3 for j in range(1, per)]
print(f"{per} kernels per replay; first start -> last end {span:.1f} us; sum of durations {dur:.1f} us; sum of gaps {sum(gaps):.1f} us "
      f"(median {sorted(gaps)[len(gaps) // 2]:.2f}, max {max(gaps):.2f})")
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for j, (nm, s, e, *_) in enumerate(f):
    k = nm.split('(')[0][:60]
    agg[k][0] += 1
    agg[k][1] += (e - s) / 1e3
    agg[k][2] += gaps[j - 1] if j else 0.0
print(f"{'kernel':60s}    n   dur us   avg   gap-before us")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:60s} {v[0]:4d} {v[1]:8.1f} {v[1] / v[0]:6.2f} {v[2]:8.1f}")
if "--all" in sys.argv:
    t0 = f[0][1]
    for j, (nm, s, e, gx, gy, gz, wx) in enumerate(f):
        print(f"{j:3d} {(s - t0) / 1e3:8.1f} {(e - s) / 1e3:6.2f} gap {gaps[j - 1] if j else 0.0:5.2f} {nm.split('(')[0][:44]:44s} grid {gx // wx}x{gy}x{gz}")
