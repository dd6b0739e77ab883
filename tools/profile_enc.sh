# The encode half's profile set for profiles/ (bash tools/round.sh encprof): per-kernel times (rocprofv3 --kernel-trace --stats over the headline workload) and HBM bytes per kernel
# (FETCH_SIZE and WRITE_SIZE in separate --pmc passes, never combined with other trace domains).  Every pass is bounded and summarised at
# once; the databases are deleted (they exceed what gpurun copies back).  Usage on the GPU box: bash tools/profile_enc.sh [label] [bench args, e.g. --slices 576 --batch 40]
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
L=${1:-r03}; shift
O=$R/gpurun_out/prof
mkdir -p $O
cd $R
timeout ${PASS_TIMEOUT:-420} rocprofv3 --kernel-trace --stats -d $O/stats -o p -- python bench.py --steps 3 --warmup 1 --legs "" --no-verify "$@" > $O/${L}_bench_under_rocprof.json 2> $O/stats.log || tail -3 $O/stats.log
db=$(find $O/stats -name "*.db" | head -1)
[ -n "$db" ] && python3 tools/rocprof_summary.py stats $db > $O/${L}_kernel_stats.csv
rm -rf $O/stats
for c in FETCH_SIZE WRITE_SIZE; do
  lc=$(echo $c | tr A-Z a-z)
  timeout ${PASS_TIMEOUT:-420} rocprofv3 --kernel-trace --pmc $c -d $O/pmc -o p -- python bench.py --steps 1 --warmup 0 --legs "" --no-verify "$@" > $O/pmc.log 2>&1 || tail -3 $O/pmc.log
  db=$(find $O/pmc -name "*.db" | head -1)
  [ -n "$db" ] && python3 tools/rocprof_summary.py pmc $db $c > $O/${L}_pmc_$lc.csv
  rm -rf $O/pmc
done
ls -la $O; cat $O/${L}_kernel_stats.csv $O/${L}_pmc_fetch_size.csv $O/${L}_pmc_write_size.csv
