# Kernel timeline of one bench step (start/end of every launch, gaps between k_resolve launches).  GPU box: bash tools/timeline.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/tl
rocprofv3 --kernel-trace -d gpurun_out/tl -o tl -- python bench.py --steps 1 --warmup 1 --legs "" --no-verify > gpurun_out/tl/log 2>&1
python - "$(find gpurun_out/tl -name '*.db' | head -1)" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in tabs else [t for t in tabs if "kernel" in t.lower()][0]
cols = [r[1] for r in db.execute(f"pragma table_info({view})")]
rows = db.execute(f"select name, start, end from {view} order by start").fetchall()
rows = [(n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip(), s, e) for n, s, e in rows]
rows = [r for r in rows if r[0].startswith("k_")]
# last step = from the last k_model launch on
starts = [i for i, r in enumerate(rows) if r[0] == "k_model"]
step = rows[starts[-1]:]
t0 = step[0][1]
print("step kernels:", len(step), "span %.1f ms" % ((max(e for _, _, e in step) - t0) / 1e6))
res = [(s, e) for n, s, e in step if n.startswith("k_resolve")]
rc = [(s, e) for n, s, e in step if n == "k_rangecode"]
print("first k_resolve starts at %.1f ms, last ends at %.1f ms; sum of durations %.1f ms; gaps between launches %.1f ms" % ((res[0][0] - t0) / 1e6, (res[-1][1] - t0) / 1e6, sum(e - s for s, e in res) / 1e6, sum(max(0, res[i + 1][0] - res[i][1]) for i in range(len(res) - 1)) / 1e6))
print("first k_rangecode starts at %.1f ms, last ends at %.1f ms; sum %.1f ms; gaps %.1f ms" % ((rc[0][0] - t0) / 1e6, (rc[-1][1] - t0) / 1e6, sum(e - s for s, e in rc) / 1e6, sum(max(0, rc[i + 1][0] - rc[i][1]) for i in range(len(rc) - 1)) / 1e6))
for n, s, e in step[-4:]: print("  tail: %-12s %.1f -> %.1f ms" % (n, (s - t0) / 1e6, (e - t0) / 1e6))
km = [(n, s, e) for n, s, e in step if n == "k_model"]
print("k_model: first start 0.0, last end %.1f ms, busy %.1f ms" % ((km[-1][2] - t0) / 1e6, sum(e - s for _, s, e in km) / 1e6))
PY
rm -rf gpurun_out/tl
