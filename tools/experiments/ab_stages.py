"""Same-box A/B of the stage timings (H3_LIB = variant library in tools/experiments/ab/): CCDM forward @128^3, AE decode / cond-encode @512^2."""
import os, sys, subprocess
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
libs = sys.argv[1:] or [""]
for rnd in range(2):
    for l in libs:
        env = dict(os.environ, H3_LIB=l, GG_PROBE_ROUNDS="4", GG_PROBE_GRAPH="1")
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "perf_probe.py"), "ccdm128", "ae"], env=env, capture_output=True, text=True).stdout
        print(f"== {l or 'product'}: " + " | ".join(x.strip() for x in out.splitlines() if "forward" in x or "AE" in x), flush=True)
