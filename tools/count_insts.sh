# Deterministic progress metric for k_resolve / k_rangecode: dynamic instruction counts (SQ_INSTS_*) per 64-symbol chunk and per
# decision at batch 64.  Usage on the GPU box: bash tools/count_insts.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/ci
RCGPU_BENCH_BATCH=64 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -d gpurun_out/ci -o ci -- python bench.py --steps 1 --warmup 0 --legs "" --no-verify > gpurun_out/ci/log 2>&1
python - "$(find gpurun_out/ci -name '*.db' | head -1)" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
ch = 64 * 64 * 414720 / 64; dec = 64 * 64 * 8.64e6 / 64
for k, c, n, s in db.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection where kernel_name like '%k_resolve%' or kernel_name like '%k_rangecode%' group by kernel_name, counter_name"):
    if "resolve" in k: print("k_resolve   %-20s %8.1f per chunk" % (c, s / ch))
    else: print("k_rangecode %-20s %8.2f per decision" % (c, s / dec))
PY
rm -rf gpurun_out/ci
