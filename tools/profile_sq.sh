# SQ issue/stall counters of the two dominant kernels (one PMC pass per group of <= 8 SQ counters; never combined with trace domains
# other than --kernel-trace).  Usage on the GPU box: bash tools/profile_sq.sh [batch]
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B=${1:-64}
EXTRA=${2:-}
mkdir -p $R/gpurun_out/sq
cd $R
export RCGPU_BENCH_BATCH=$B
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d gpurun_out/sq/a -o sq -- python bench.py --steps 1 --warmup 0 --legs "" --no-verify $EXTRA > gpurun_out/sq/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES -d gpurun_out/sq/b -o sq -- python bench.py --steps 1 --warmup 0 --legs "" --no-verify $EXTRA > gpurun_out/sq/b.log 2>&1
for d in a b; do
  db=$(find gpurun_out/sq/$d -name "*.db" | head -1)
  python - "$db" <<'PY' > gpurun_out/sq/$d.csv
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name").fetchall()
print("kernel,counter,launches,sum")
for k, c, n, s in rows:
    k = k.replace("(anonymous namespace)::", "").split("(")[0].strip()
    if k.startswith("k_"): print(f"{k},{c},{n},{s:.0f}")
PY
done
cat gpurun_out/sq/a.csv gpurun_out/sq/b.csv
