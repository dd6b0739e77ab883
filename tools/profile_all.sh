# Everything profiles/ holds for a round, in one call on the GPU box; only small CSV summaries are left under gpurun_out/summary
# (gpurun copies back at most 64 MiB).  Usage: bash tools/profile_all.sh <tag>   e.g. r01c
TAG=${1:-r01c}
R=$GRAFT_REPO_ROOT
cd $R
bash tools/profile_round.sh > /dev/null 2>&1
mkdir -p gpurun_out/summary
python tools/rocprof_summary.py stats $(find gpurun_out/prof/stats -name "*.db" | head -1) > gpurun_out/summary/${TAG}_kernel_stats.csv
python tools/rocprof_summary.py pmc $(find gpurun_out/prof/pmc_fetch -name "*.db" | head -1) FETCH_SIZE > gpurun_out/summary/${TAG}_pmc_fetch_size.csv
python tools/rocprof_summary.py pmc $(find gpurun_out/prof/pmc_write -name "*.db" | head -1) WRITE_SIZE > gpurun_out/summary/${TAG}_pmc_write_size.csv
grep -o '{"metric.*' gpurun_out/prof/bench_stats.log | head -1 > gpurun_out/summary/${TAG}_bench_under_rocprof.json
echo "run,kernel,counter,launches,sum" > gpurun_out/summary/${TAG}_sq_counters.csv
for mode in encode check; do
  if [ $mode = check ]; then bash tools/profile_sq.sh 64 "--mode check" > /dev/null 2>&1; else bash tools/profile_sq.sh 64 > /dev/null 2>&1; fi
  for d in a b; do
    python - "$(find gpurun_out/sq/$d -name '*.db' | head -1)" $mode <<'PY' >> gpurun_out/summary/${TAG}_sq_counters.csv
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
keep = ("k_resolve<false>", "k_rangecode") if sys.argv[2] == "encode" else ("k_dec_slices<true>", "k_dec_slices<false>", "k_dec_slices")
for k, c, n, s in db.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"):
    k = k.replace("(anonymous namespace)::", "").split("(")[0].strip()
    if k.startswith("void "): k = k[5:]
    if k in keep: print(f"{sys.argv[2]},{k},{c},{n},{s:.0f}")
PY
  done
  rm -rf gpurun_out/sq
done
rm -rf gpurun_out/prof
ls -la gpurun_out/summary
