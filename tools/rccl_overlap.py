"""Where do the gradient all-reduce kernels sit in a traced step?  For every RCCL kernel of the last steps of a packed
trace (tools/trace_pack.py): its start relative to the step, its duration, and how much of it ran under other kernels.
usage: rccl_overlap.py trace.csv.gz [steps=4]"""
import gzip
import sys

path = sys.argv[1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
with gzip.open(path, "rt") as f:
    n = int(f.readline())
    names = {}
    for _ in range(n):
        k, nm = f.readline().rstrip("\n").split("\t", 1)
        names[int(k)] = nm
    rows = sorted((tuple(int(v) for v in line.split(",")) for line in f), key=lambda r: r[1])
adam = {k for k, nm in names.items() if "FusedOptimizerTensorListMetadata" in nm}
rccl = {k for k, nm in names.items() if "nccl" in nm.lower() or "rccl" in nm.lower()}
if not rccl:
    # measured on the 1-GPU boxes of this environment: with world size 1 the all-reduce launches NO RCCL kernel at all
    # (the collective degenerates to nothing), so the overlap of the in-backward bucket launch can only be traced on a
    # >= 2-GPU node
    print("no RCCL kernel in this trace (a 1-rank all-reduce launches none)")
    sys.exit(0)
ad = [r for r in rows if r[0] in adam]
groups, cur = [], [ad[0]]
for r in ad[1:]:
    if r[1] - cur[-1][2] > 5_000_000:
        groups.append(cur)
        cur = [r]
    else:
        cur.append(r)
groups.append(cur)
ends = [g[-1][2] for g in groups][-(2 * nsteps + 1):]
bounds = ends[::2]
print("RCCL kernel names:", sorted({names[k][:70] for k in rccl}))
for si, (a, b) in enumerate(zip(bounds[:-1], bounds[1:])):
    rs = [r for r in rows if a < r[2] <= b]
    coll = [r for r in rs if r[0] in rccl]
    others = [r for r in rs if r[0] not in rccl]
    tot = sum(e - s for _, s, e in coll)
    under = 0
    for _, s, e in coll:
        covered, cur_e = 0, s
        for _, os_, oe in others:
            if oe <= cur_e or os_ >= e:
                continue
            lo, hi = max(os_, cur_e), min(oe, e)
            if hi > lo:
                covered += hi - lo
                cur_e = hi
        under += covered
    print("step %d: wall %.1f ms, %d collective kernels, %.2f ms in total, %.2f ms (%.0f %%) of it concurrent with compute kernels"
          % (si, (b - a) / 1e6, len(coll), tot / 1e6, under / 1e6, 100.0 * under / max(tot, 1)))
    for _, s, e in coll[:12]:
        print("    +%.1f ms  %.3f ms" % ((s - a) / 1e6, (e - s) / 1e6))
