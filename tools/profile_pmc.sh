# Hardware counters of the encoder's kernels, one rocprofv3 --pmc pass per group (never combined with trace domains other than
# --kernel-trace; under counter collection the kernels of a step run one after the other).  Usage on the GPU box:
#   [RCGPU_BENCH_BATCH=64] [PMC_GROUPS="1 2 3"] bash tools/profile_pmc.sh <label> [bench args...]      (RCGPU_RC_SPAN etc. are inherited)
# writes gpurun_out/pmc/<label>.csv : kernel,counter,launches,sum.  Every pass is bounded (PMC_PASS_TIMEOUT, default 240 s) and its
# database is deleted once summarised: a pass over a 336-frame step took more than the box allows, and the databases exceed what
# gpurun copies back.
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
L=${1:-run}; shift
mkdir -p $R/gpurun_out/pmc/$L
cd $R
export RCGPU_BENCH_BATCH=${RCGPU_BENCH_BATCH:-64}
summ() {
python3 - "$1" <<'PY'
import glob, sqlite3, sys
for db in sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True)):
    c = sqlite3.connect(db)
    try:
        rows = c.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name").fetchall()
    except Exception as e:
        print("#", db, e); continue
    for k, n, cnt, s in rows:
        k = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()
        if k.startswith("k_"): print(f"{k},{n},{cnt},{s:.0f}")
PY
}
echo "kernel,counter,launches,sum" > gpurun_out/pmc/$L.csv
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  case " ${PMC_GROUPS:-1 2 3 4 5 6 7} " in *" $i "*) ;; *) continue;; esac
  t0=$(date +%s)
  timeout ${PMC_PASS_TIMEOUT:-240} rocprofv3 --kernel-trace --pmc $group -d gpurun_out/pmc/$L/p$i -o p -- python bench.py --steps 1 --warmup 0 --legs "" --no-verify "$@" > gpurun_out/pmc/$L/p$i.log 2>&1 || { echo "# pass $i failed or timed out: $group"; tail -2 gpurun_out/pmc/$L/p$i.log; }
  summ gpurun_out/pmc/$L/p$i >> gpurun_out/pmc/$L.csv
  rm -rf gpurun_out/pmc/$L/p$i
  echo "# pass $i: $(( $(date +%s) - t0 )) s"
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES
SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_MISSES SQC_ICACHE_REQ SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL
TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
GROUPS
rm -rf $R/gpurun_out/pmc/$L
grep -E "k_resolve|k_rangecode|k_rc_range|^#" $R/gpurun_out/pmc/$L.csv
