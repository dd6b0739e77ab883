# Copy-engine timeline of the host_pipeline leg: per direction the copies' durations, rates and the gaps between them, batch by batch.
# GPU box: bash tools/timeline_dma.sh   (kernel + memory-copy trace, no counters)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/tld
rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/tld -o tld -- python bench.py --steps 1 --warmup 1 --legs host --host-frames 1680 --no-verify > gpurun_out/tld/log 2>&1
tail -c 600 gpurun_out/tld/log
python - "$(find gpurun_out/tld -name '*.db' | head -1)" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
mc = [t for t in tabs if "memory_cop" in t.lower()]
print("tables:", mc)
view = "memory_copies" if "memory_copies" in tabs else mc[0]
cols = [r[1] for r in db.execute(f"pragma table_info({view})")]
print(cols)
rows = db.execute(f"select * from {view} order by start").fetchall()
ix = {c: i for i, c in enumerate(cols)}
def g(r, *names):
    for n in names:
        if n in ix: return r[ix[n]]
big = [r for r in rows if (g(r, "size", "bytes") or 0) > (8 << 20)]
print(len(rows), "copies,", len(big), "larger than 8 MB")
kv = "kernels" if "kernels" in tabs else [t for t in tabs if "kernel" in t.lower()][0]
kern = [(n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip().split("<")[0], s_, e_) for n, s_, e_ in db.execute(f"select name, start, end from {kv} order by start")]
km = [s_ for n, s_, e_ in kern if n == "k_model"]
kinds = {}
for r in big: kinds.setdefault(str(g(r, "name")), []).append(r)
print({k: len(v) for k, v in kinds.items()})
def bursts(sel, gap_ms=20.0):
    out = []; cur = None
    for r in sel:
        s_, e_, z = g(r, "start"), g(r, "end"), g(r, "size")
        if cur is None or (s_ - cur[1]) / 1e6 > gap_ms:
            if cur: out.append(cur)
            cur = [s_, e_, z, 1]
        else: cur[1] = max(cur[1], e_); cur[2] += z; cur[3] += 1
    if cur: out.append(cur)
    return out
t00 = km[-5] if len(km) >= 5 else km[0]
ev = []
for k, v in kinds.items():
    for b_ in bursts([r for r in v if g(r, "start") >= t00 - 2e9]):
        ev.append((b_[0], "%s burst: %d copies, %.1f GB, %.3f s = %.1f GB/s" % (k.replace("MEMORY_COPY_", ""), b_[3], b_[2] / 1e9, (b_[1] - b_[0]) / 1e9, b_[2] / max(1, b_[1] - b_[0])), b_[1]))
for i, t in enumerate(km[-5:]):
    st = [x for x in kern if x[1] >= t and (i == 4 or x[1] < km[-5:][i + 1] if i < 4 else True)]
    res = [x for x in st if x[0] == "k_resolve"]; ga = [x for x in st if x[0] == "k_gather"]; mo = [x for x in st if x[0] == "k_model"]
    ev.append((mo[0][1], "k_model of batch %d" % i, mo[0][2]))
    if res: ev.append((res[0][1], "k_resolve x%d of batch %d (sum %.0f ms)" % (len(res), i, sum(e_ - s_ for _, s_, e_ in res) / 1e6), res[-1][2]))
    for x in ga: ev.append((x[1], "k_gather (batch %d's stream position)" % i, x[2]))
for s_, what, e_ in sorted(ev):
    if s_ >= t00 - 1e9: print("%9.1f ms .. %9.1f ms  %s" % ((s_ - t00) / 1e6, (e_ - t00) / 1e6, what))
PY
rm -rf gpurun_out/tld
