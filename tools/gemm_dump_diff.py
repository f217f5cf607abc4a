"""Compares two per-launch GEMM tables written by bench.py under DGQ_BENCH_GEMM_DUMP (same model, two kernel plans): time per
shape class, sorted by the difference.   usage: python tools/gemm_dump_diff.py a.txt b.txt"""
import sys
from collections import defaultdict


def load(p):
    d = defaultdict(lambda: [0, 0.0])
    for line in open(p):
        key, us = line.rsplit(" ", 1)
        d[key][0] += 1
        d[key][1] += float(us)
    return d


a, b = load(sys.argv[1]), load(sys.argv[2])
rows = []
for k in a:
    if k in b:
        rows.append((b[k][1] - a[k][1], k, a[k][0], a[k][1], b[k][1]))
rows.sort()
print("%-60s %5s %10s %10s %9s" % ("M,N,K,Kp,mode,out_bytes", "n", "A us", "B us", "B-A us"))
for dlt, k, n, ta, tb in rows:
    print("%-60s %5d %10.1f %10.1f %+9.1f" % (k[:60], n, ta, tb, dlt))
print("total A %.1f us   B %.1f us" % (sum(v[1] for v in a.values()), sum(v[1] for v in b.values())))
