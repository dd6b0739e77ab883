# the check half as the driver's line runs it, a few times.  bash tools/r04_c64.sh [n]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
for k in $(seq 1 ${1:-2}); do
  timeout 600 python bench.py --mode check --steps 2 --warmup 1 --legs "" > gpurun_out/r04/c64_$k.json 2> gpurun_out/r04/c64_$k.err || tail -3 gpurun_out/r04/c64_$k.err
  python - $k <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r04/c64_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("check", d["value"], "frames/s", d["ms_per_step"], "ms", d["roofline"]["kernel_ms"], d["roofline"]["request_frac"], d["config"]["all_frames_identical_to_source"], d["config"]["md5_matches_hashlib"])
PY
done
