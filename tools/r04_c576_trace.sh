# kernel trace of the check half at 576 slices: every launch's start and duration.  bash tools/r04_c576_trace.sh <check batch>
CFG="$1"; B=$2
cd /tmp && export TMPDIR=/tmp
export RCGPU_NO_CU_PARTITION=1      # rocprofv3 of ROCm 7.2 crashes in a process with CU-masked streams (the hash then shares the CUs again)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/pc gpurun_out/r04
timeout 420 rocprofv3 --kernel-trace -d gpurun_out/pc -o chk -- python bench.py --mode check --steps ${STEPS:-2} --warmup 1 --legs "" $CFG > gpurun_out/pc/log 2>&1
python - "$(find gpurun_out/pc -name '*.db' | head -1)" <<'PY' | tee gpurun_out/r04/c576_trace_$B.txt
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if t.startswith("kernels") or t == "kernels"]
rows = list(db.execute("select name, start, end from kernels order by start"))
t0 = rows[0][1]
for n, s, e in rows:
    if any(k in n for k in ("k_dec", "k_md5", "k_compare", "k_pack")) or (e - s) > 5e6:
        print("%-40s start %9.3f ms  dur %9.3f ms" % (n.split("(")[0][-40:], (s - t0) / 1e6, (e - s) / 1e6))
PY
rm -rf gpurun_out/pc
