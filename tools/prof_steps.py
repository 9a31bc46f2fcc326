"""Split the tail of a packed kernel trace (tools/trace_pack.py) into train() steps and report, per step, the wall
time, the UNION of the kernel intervals (time with at least one kernel running), the sum of the kernel durations, and
the kernels whose time differs most between gradient-penalty steps (every 4th) and ordinary steps.
A step boundary is the last launch of the fused Adam kernel of the generator phase (the G optimiser runs last in train()).
usage: prof_steps.py trace.csv.gz [steps=8] [top=40]"""
import gzip
import sys

path = sys.argv[1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
with gzip.open(path, "rt") as f:
    n = int(f.readline())
    names = {}
    for _ in range(n):
        k, nm = f.readline().rstrip("\n").split("\t", 1)
        names[int(k)] = nm
    rows = sorted(tuple(int(v) for v in line.split(",")[:3]) for line in f)  # (a 4th column, the stream id, is ignored here)
rows.sort(key=lambda r: r[1])
adam = [k for k, nm in names.items() if ("FusedOptimizerTensorListMetadata" in nm or "adam_pack_kernel" in nm)]  # the fused Adam only (other multi-tensor ops exist)
ad = [r for r in rows if r[0] in adam]
# groups of Adam launches separated by > 5 ms: one group per optimiser step (D, G, D, G, ...)
groups, cur = [], [ad[0]]
for r in ad[1:]:
    if r[1] - cur[-1][2] > 5_000_000:
        groups.append(cur)
        cur = [r]
    else:
        cur.append(r)
groups.append(cur)
ends = [g[-1][2] for g in groups]
# a train() step = two optimiser groups; take the last 2*nsteps groups, boundaries at every second group end
ends = ends[-(2 * nsteps + 1):]
bounds = ends[::2]
steps = []
for a, b in zip(bounds[:-1], bounds[1:]):
    rs = [r for r in rows if a < r[2] <= b]
    union, cur_e, gaps = 0, a, 0
    for _, s, e in rs:
        if s > cur_e:
            gaps += s - cur_e
            cur_e = e
            union += e - s
        elif e > cur_e:
            union += e - cur_e
            cur_e = e
    agg = {}
    for k, s, e in rs:
        x = agg.setdefault(k, [0, 0])
        x[0] += 1
        x[1] += e - s
    steps.append(dict(wall=(b - a) / 1e6, union=union / 1e6, sum=sum(e - s for _, s, e in rs) / 1e6, n=len(rs), agg=agg))
for i, s in enumerate(steps):
    print("step %d: wall %.1f ms  union(busy) %.1f ms  idle %.1f ms  sum %.1f ms  kernels %d" % (i, s["wall"], s["union"], s["wall"] - s["union"], s["sum"], s["n"]))
big = sorted(range(len(steps)), key=lambda i: -steps[i]["sum"])
ngp = max(1, len(steps) // 4)
gp, plain = big[:ngp], big[ngp:]
print("gradient-penalty steps (by kernel time): %s" % sorted(gp))


def mean(idx, k, j):
    return sum(steps[i]["agg"].get(k, [0, 0])[j] for i in idx) / len(idx)


diff = []
for k in names:
    d = mean(gp, k, 1) - mean(plain, k, 1)
    if d:
        diff.append((d, k))
print("%9s %9s %9s %8s %8s  kernel" % ("d ms", "gp ms", "plain ms", "gp n", "plain n"))
for d, k in sorted(diff, key=lambda x: -abs(x[0]))[:top]:
    print("%9.3f %9.3f %9.3f %8.1f %8.1f  %s" % (d / 1e6, mean(gp, k, 1) / 1e6, mean(plain, k, 1) / 1e6, mean(gp, k, 0), mean(plain, k, 0), names[k][:110]))

if len(sys.argv) > 4:  # per-family totals of the ordinary steps
    fam = {}
    for k in names:
        t, c = mean(plain, k, 1) / 1e6, mean(plain, k, 0)
        if not c:
            continue
        nm = names[k]
        if "anonymous namespace)::" in nm and "at::native" not in nm or nm.startswith("bias_partial"):
            f = "ours"
        elif "miopen" in nm.lower() or "igemm_" in nm or "MIOpen" in nm or "gtc" in nm:
            f = "miopen"
        elif nm.startswith("Cijk"):
            f = "hipblaslt"
        else:
            f = "torch"
        a = fam.setdefault(f, [0.0, 0.0, []])
        a[0] += t
        a[1] += c
        a[2].append((t, c, nm))
    for f, (t, c, lst) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
        print("== %s: %.2f ms/step, %.0f launches" % (f, t, c))
        if f in sys.argv[4].split(","):
            for t, c, nm in sorted(lst, reverse=True)[:int(sys.argv[5]) if len(sys.argv) > 5 else 30]:
                print("   %7.3f %6.1f  %s" % (t, c, nm[:150]))
