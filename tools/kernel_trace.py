#!/usr/bin/env python3
"""A rocprofv3 --kernel-trace database as a timeline: every long kernel's start and duration, and -- for the encoder's run-on steps -- where the
k_resolve and k_rangecode streams stand still.   python tools/kernel_trace.py <results.db> [encode|check]   (bash tools/round.sh trace <tag> ...)"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
mode = sys.argv[2] if len(sys.argv) > 2 else "encode"
rows = [(n or "", s, e) for n, s, e in db.execute("select name, start, end from kernels order by start")]
t0 = rows[0][1]
if mode == "check":
    for n, s, e in rows:
        if any(k in n for k in ("k_dec", "k_md5", "k_compare", "k_pack")) or (e - s) > 5e6:
            print("%-40s start %9.3f ms  dur %9.3f ms" % (n.split("(")[0][-40:], (s - t0) / 1e6, (e - s) / 1e6))
    sys.exit(0)


def of(key):
    return [(s - t0, e - t0) for n, s, e in rows if key in n]


res, rc, mod, gat = of("k_resolve"), of("k_rangecode"), of("k_model"), of("k_gather")
print("launches: resolve %d, rangecode %d, model %d" % (len(res), len(rc), len(mod)))
for name, iv in (("k_resolve", res), ("k_rangecode", rc)):
    for a, b in zip(iv, iv[1:]):
        g = (b[0] - a[1]) / 1e6
        if g > 1.0:
            print("  %s idle %.1f ms at %.1f ms" % (name, g, a[1] / 1e6))
for m in mod:
    print("  k_model %.1f .. %.1f ms" % (m[0] / 1e6, m[1] / 1e6))
for g in gat:
    print("  k_gather %.1f .. %.1f ms" % (g[0] / 1e6, g[1] / 1e6))
n = 32
for b in range(len(res) // n):
    r = res[b * n:(b + 1) * n]
    c = rc[b * n:(b + 1) * n]
    if r and c:
        print("batch %d: resolve %.1f .. %.1f (busy %.1f), coder %.1f .. %.1f (busy %.1f)" % (b, r[0][0] / 1e6, r[-1][1] / 1e6, sum(e - s for s, e in r) / 1e6,
                                                                                                c[0][0] / 1e6, c[-1][1] / 1e6, sum(e - s for s, e in c) / 1e6))
