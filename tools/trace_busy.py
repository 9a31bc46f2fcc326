"""Per-step GPU occupancy from a packed kernel trace (tools/trace_pack.py): wall, sum of kernel durations, union of busy
intervals, idle time, time at concurrency 1 / 2 / 3+, per-stream kernel time, and the kernels that follow the longest idle
gaps.  Steps are delimited by the adam_pack launches (two per train() call).
usage: trace_busy.py trace.csv.gz [first_step] [n_steps]"""
import collections
import gzip
import sys


def load(path):
    names, rows = {}, []
    for line in gzip.open(path, "rt"):
        line = line.rstrip("\n")
        if "\t" in line:
            i, n = line.split("\t", 1)
            names[int(i)] = n
        elif "," in line:
            a = line.split(",")
            rows.append((int(a[1]), int(a[2]), int(a[0]), a[3] if len(a) > 3 else "0"))
    rows.sort()
    return names, rows


def main():
    names, rows = load(sys.argv[1])
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    nst = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    adam = {i for i, n in names.items() if "adam_pack" in n}
    marks = [r[0] for r in rows if r[2] in adam]
    t0, t1 = marks[2 * first], marks[2 * (first + nst)]
    sel = [r for r in rows if t0 <= r[0] < t1]
    tot = sum(e - s for s, e, _, _ in sel)
    busy, gaps = 0, []
    cs, ce = sel[0][0], sel[0][1]
    for s, e, i, q in sel[1:]:
        if s > ce:
            busy += ce - cs
            gaps.append((s - ce, i))
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    ms = lambda v: v / 1e6 / nst
    print("per step over %d steps: wall %.2f ms, kernel-sum %.2f, union-busy %.2f, idle %.2f, launches %.0f" % (
        nst, ms(t1 - t0), ms(tot), ms(busy), ms(t1 - t0 - busy), len(sel) / nst))
    ev = []
    for s, e, _, _ in sel:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    c, last, hist = 0, ev[0][0], collections.Counter()
    for t, d in ev:
        hist[min(c, 3)] += t - last
        last, c = t, c + d
    print("concurrency 0/1/2/3+: " + " / ".join("%.2f" % ms(hist[k]) for k in range(4)) + " ms per step")
    per_q = collections.Counter()
    for s, e, _, q in sel:
        per_q[q] += e - s
    print("kernel time per stream: " + ", ".join("%s: %.2f" % (q, ms(v)) for q, v in per_q.most_common(8)))
    by = collections.Counter()
    for g, i in gaps:
        by[names[i][:90]] += g
    print("idle time by the kernel that ends the gap (ms per step):")
    for n, v in by.most_common(12):
        print("  %.3f  %s" % (ms(v), n))


if __name__ == "__main__":
    main()
