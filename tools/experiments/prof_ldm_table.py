"""Per-kernel table of a rocprofv3 --kernel-trace database of ab_ldm_forward.py (306 forwards: 1 eager + capture + 5 + 300 replays)."""
import sqlite3, re, sys, subprocess
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
q = f"select s.kernel_name, count(*), avg(d.end-d.start)/1000.0, min(d.end-d.start)/1000.0, d.grid_size_x, d.workgroup_size_x from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name, d.grid_size_x order by 3*count(*) desc"
rows = [r for r in db.execute(q) if r[1] >= 300]
tot = 0.0
fam = {}
for r in rows:
    n = re.sub(r'^_Z\d+', '', r[0]); n = re.sub(r'(Ev|PK|10ConvParams|10AttnParams).*$', '', n)[:46]
    per = r[1] / 306.0
    tot += per * r[2]
    key = re.sub(r'ILi.*', '', n)
    fam[key] = fam.get(key, [0, 0.0]); fam[key][0] += per; fam[key][1] += per * r[2]
    print(f"{n:46s} x{per:5.1f} avg {r[2]:6.2f} min {r[3]:6.2f} wgs {r[4]//r[5]:4d}")
print(f"sum of durations per forward: {tot:.1f} us")
for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1]): print(f"  {k:32s} {v[0]:6.1f} launches {v[1]:7.1f} us")
