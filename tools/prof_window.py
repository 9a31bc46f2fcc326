"""Summarise the LAST `window_ms` of a packed kernel trace (tools/trace_pack.py): steady-state steps only, so
library auto-tuning during warm-up does not pollute the picture.
usage: prof_window.py trace.csv.gz window_ms steps [top]"""
import gzip
import sys

path, window_ms, steps = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
top = int(sys.argv[4]) if len(sys.argv) > 4 else 40
with gzip.open(path, "rt") as f:
    n = int(f.readline())
    names = {}
    for _ in range(n):
        k, nm = f.readline().rstrip("\n").split("\t", 1)
        names[int(k)] = nm
    rows = [tuple(int(v) for v in line.split(",")[:3]) for line in f]  # (a 4th column, the stream id, is ignored here)
t_end = max(r[2] for r in rows)
t0 = t_end - int(window_ms * 1e6)
agg = {}
busy = 0
for k, s, e in rows:
    if s < t0:
        continue
    a = agg.setdefault(k, [0, 0])
    a[0] += 1
    a[1] += e - s
    busy += e - s
print("window %.1f ms, %g steps: kernel-busy %.1f ms/step (%.0f%% of wall)" % (window_ms, steps, busy / 1e6 / steps,
                                                                           100 * busy / 1e6 / window_ms))
print("%9s %8s %9s  %s" % ("ms/step", "calls/st", "avg us", "kernel"))
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%9.3f %8.1f %9.1f  %s" % (t / 1e6 / steps, c / steps, t / 1e3 / c, names[k][:130]))
