"""Join the conv shapes of one latent-UNet forward (ldm_conv_list.py) with the per-kernel durations of a captured replay
(ldm_graph_timeline.py --all): python tools/experiments/ldm_conv_table.py list.txt timeline.txt"""
import collections, re, sys
convs = [l.strip() for l in open(sys.argv[1]) if "->" in l]
tl = []
for l in open(sys.argv[2]):
    m = re.match(r"^\s*(\d+)\s+([\d.]+)\s+([\d.]+) gap\s+([\d.-]+) (.*?)\s+grid (\S+)", l)
    if m:
        tl.append((float(m.group(3)), m.group(5).strip(), m.group(6)))
ck = [t for t in tl if any(s in t[1] for s in ("conv_box2d", "conv_gather5", "conv_tinym", "conv_gather_kernel"))]
assert len(convs) == len(ck), (len(convs), len(ck))
agg = collections.OrderedDict()
for c, (d, nm, g) in zip(convs, ck):
    a = agg.setdefault((c, nm[:42], g), [0, 0.0])
    a[0] += 1
    a[1] += d
print(f"{len(ck)} conv launches, {sum(t[0] for t in ck):.1f} us of the replay's {sum(t[0] for t in tl):.1f} us (rocprofv3 durations of a captured replay: back to back, +~0.7 us per kernel of tool overhead)")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{v[1]:7.1f} us n={v[0]:2d} avg {v[1] / v[0]:5.2f}  {k[0]:40s} {k[1]:42s} {k[2]}")
