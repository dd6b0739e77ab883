# Kernel timeline of the host_pipeline leg: per batch, the span of its kernels and the idle time between consecutive batches and
# between k_resolve launches.  GPU box: bash tools/timeline_pipe.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/tlp
rocprofv3 --kernel-trace -d gpurun_out/tlp -o tlp -- python bench.py --steps 1 --warmup 1 --legs host --host-frames 1920 --no-verify > gpurun_out/tlp/log 2>&1
tail -c 1500 gpurun_out/tlp/log
python - "$(find gpurun_out/tlp -name '*.db' | head -1)" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in tabs else [t for t in tabs if "kernel" in t.lower()][0]
rows = db.execute(f"select name, start, end from {view} order by start").fetchall()
rows = [(n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip(), s, e) for n, s, e in rows]
rows = [r for r in rows if r[0].startswith("k_")]
starts = [i for i, r in enumerate(rows) if r[0] == "k_model"]
# the pipeline leg = the last 5 k_model launches (1920 frames / the batch)
starts = starts[-5:]
t00 = rows[starts[0]][1]
prev_end = None
for bi, st in enumerate(starts):
    en = starts[bi + 1] if bi + 1 < len(starts) else len(rows)
    step = rows[st:en]
    t0 = step[0][1]
    res = [(s, e) for n, s, e in step if n.startswith("k_resolve")]
    rc = [(s, e) for n, s, e in step if n == "k_rangecode"]
    last = max(e for _, _, e in step)
    gap_prev = (t0 - prev_end) / 1e6 if prev_end else 0
    print("batch %d: k_model starts %.1f ms (idle since previous batch's last kernel %.1f ms), k_model %.1f ms, first k_resolve +%.1f ms, resolve sum %.1f gaps %.1f, rangecode sum %.1f gaps %.1f, last kernel ends +%.1f ms"
          % (bi, (t0 - t00) / 1e6, gap_prev, (step[0][2] - t0) / 1e6, (res[0][0] - t0) / 1e6, sum(e - s for s, e in res) / 1e6,
             sum(max(0, res[i + 1][0] - res[i][1]) for i in range(len(res) - 1)) / 1e6, sum(e - s for s, e in rc) / 1e6,
             sum(max(0, rc[i + 1][0] - rc[i][1]) for i in range(len(rc) - 1)) / 1e6, (last - t0) / 1e6))
    for n, s, e in step[-3:]:
        print("      tail: %-12s +%.1f -> +%.1f ms" % (n, (s - t0) / 1e6, (e - t0) / 1e6))
    prev_end = last
mc = [t for t in tabs if "memory_cop" in t.lower() or "memcpy" in t.lower()]
print("memory copy tables:", mc)
PY
rm -rf gpurun_out/tlp
