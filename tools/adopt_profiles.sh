# After `gpurun -- 'bash tools/round.sh refresh <tag>'`: the summaries of that run become the round's profiles/ files, and profiles/traffic.json is
# rebuilt from them (round.sh refresh runs this on the box too, before its bench line, so that the line carries the fresh figures)
# -- with the sha256 of the kernel sources as they are now (tools/update_traffic.py).  Usage: bash tools/adopt_profiles.sh <tag> [round, default r05]
set -e
T=$1; R=${2:-r05}; cd "$(dirname "$0")/.."; O=gpurun_out/$R
for f in kernel_stats.csv pmc_fetch_size.csv pmc_write_size.csv bench_under_rocprof.json; do
    cp $O/${R}${T}_$f profiles/${R}_$f
    cp $O/${R}${T}_576_$f profiles/${R}_576_$f 2>/dev/null || true; cp $O/${R}${T}_8k_$f profiles/${R}_8k_$f 2>/dev/null || true
    cp $O/${R}${T}_check_$f profiles/${R}_check_$f
done
rm -f profiles/${R}_576_bench_under_rocprof.json profiles/${R}_8k_bench_under_rocprof.json
cp $O/${R}${T}_sq.csv profiles/${R}_sq_counters.csv
cp $O/check_partitioned_$T.json profiles/${R}_check_partitioned.json; cp $O/check576_partitioned_$T.json profiles/${R}_check576_partitioned.json
python tools/update_traffic.py $R
[ -s $O/bench_$T.json ] && python - $O/bench_$T.json profiles/${R}_bench_default.json <<'PY'
import json, sys
line = open(sys.argv[1]).read().strip().splitlines()[-1]
d = json.loads(line); open(sys.argv[2], "w").write(line + "\n")
print("bench line:", d["value"], d["unit"], "wall", d["config"].get("bench_wall_seconds"), "traffic_stale", d["roofline"].get("traffic_stale"))
PY
true
