"""Every (class, conv shape) of a training step with its hipEvent time: bench.py's roofline.per_layer with no row limit,
printed as a table (sorted by time; `--small` keeps the <= 16 px layers).
Usage (GPU box): python tools/layer_table.py [--small] [bench args...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
small = "--small" in sys.argv
args = [a for a in sys.argv[1:] if a != "--small"]
out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--bench-a-steps", "0", "--fp32-steps", "0",
                      "--steps", "2", "--warmup", "2", "--per-layer-top", "0"] + args, capture_output=True, text=True).stdout
line = [l for l in out.splitlines() if l.startswith("{")][-1]
j = json.loads(line)
steps = 4
rows = j["roofline"]["per_layer"]["layers"]
tot = 0.0
print("%-44s %8s %9s %8s %7s %6s" % ("layer", "launches", "us/launch", "ms/step", "TF/s", "frac"))
for r in rows:
    res = int(r["layer"].split("@")[1].split("x")[0])
    if small and res > 16:
        continue
    tot += r["ms_total"] / steps
    print("%-44s %8.1f %9.1f %8.3f %7.1f %6.3f" % (r["layer"], r["launches"] / steps, r["ms_per_launch"] * 1e3, r["ms_total"] / steps,
                                                 r["tflops"], r["frac"]))
print("total %.2f ms/step" % tot)
