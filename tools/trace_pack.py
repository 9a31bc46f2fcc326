"""Pack a rocprofv3 kernel_trace.csv into name-id/start/end/stream rows (small enough to bring back from the GPU box).
usage: trace_pack.py kernel_trace.csv out.csv.gz"""
import csv
import gzip
import sys

GRID = len(sys.argv) > 3 and sys.argv[3] == "grid"
names = {}
with open(sys.argv[1]) as f, gzip.open(sys.argv[2], "wt") as g:
    rows = []
    for r in csv.DictReader(f):
        # the grid size distinguishes the layer shapes a kernel template is launched with
        k = names.setdefault(r["Kernel_Name"] + (" grid=%s" % r.get("Grid_Size_X", "?") if GRID else ""), len(names))
        rows.append("%d,%s,%s,%s\n" % (k, r["Start_Timestamp"], r["End_Timestamp"], r.get("Stream_Id") or r.get("Queue_Id") or "0"))
    g.write("%d\n" % len(names))
    for n, k in names.items():
        g.write("%d\t%s\n" % (k, n))
    g.writelines(rows)
