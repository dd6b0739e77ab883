# kernel trace of the run-on steps: where the front stream (k_resolve) and the coder's stream (k_rangecode) stand still.  bash tools/r04_runon_trace.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/pc gpurun_out/r04
GPU_MAX_HW_QUEUES=${Q:-8} timeout 420 rocprofv3 --kernel-trace -d gpurun_out/pc -o ro -- python bench.py --steps 4 --warmup 1 --legs "" --no-verify --run-on ${RO:-1} > gpurun_out/pc/log 2>&1
python - "$(find gpurun_out/pc -name '*.db' | head -1)" <<'PY' | tee gpurun_out/r04/runon_trace.txt
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = [(n or "", s, e) for n, s, e in db.execute("select name, start, end from kernels order by start")]
t0 = rows[0][1]
def of(key): return [(s - t0, e - t0) for n, s, e in rows if key in n]
res, rc, mod, gat, foot = of("k_resolve"), of("k_rangecode"), of("k_model"), of("k_gather"), of("k_footer")
print("launches: resolve %d, rangecode %d, model %d" % (len(res), len(rc), len(mod)))
def gaps(name, iv, thr=1.0):
    for a, b in zip(iv, iv[1:]):
        g = (b[0] - a[1]) / 1e6
        if g > thr: print("  %s idle %.1f ms at %.1f ms" % (name, g, a[1] / 1e6))
gaps("k_resolve", res); gaps("k_rangecode", rc)
for m in mod: print("  k_model %.1f .. %.1f ms" % (m[0] / 1e6, m[1] / 1e6))
for g in gat: print("  k_gather %.1f .. %.1f ms" % (g[0] / 1e6, g[1] / 1e6))
n = 32
for b in range(len(res) // n):
    r = res[b * n:(b + 1) * n]; c = rc[b * n:(b + 1) * n]
    print("batch %d: resolve %.1f .. %.1f (busy %.1f), coder %.1f .. %.1f (busy %.1f)" % (b, r[0][0] / 1e6, r[-1][1] / 1e6, sum(e - s for s, e in r) / 1e6, c[0][0] / 1e6, c[-1][1] / 1e6, sum(e - s for s, e in c) / 1e6))
PY
tail -c 600 gpurun_out/pc/log | cut -c1-300; rm -rf gpurun_out/pc
