# the linked reference's --check at 576 slices per frame (1000 frames), one batch ahead and not.  bash tools/r04_l576.sh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
for ahead in ${AHEADS:-1 0}; do
  export RCGPU_CHECK_AHEAD=$ahead
  RCGPU_TRACE_KEPT=1 RCGPU_LINKED_VARIANTS=device_decoder,device_decoder_and_sources timeout 900 python bench.py --mode check --legs cpu --steps 1 --warmup 0 --slices 576 --check-batch 336 > gpurun_out/r04/l576_$ahead.json 2> gpurun_out/r04/l576_$ahead.err; tail -12 gpurun_out/r04/l576_$ahead.err | cut -c1-300
  python - $ahead <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r04/l576_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
l = d["linked_check"]
for k, v in l.items():
    if isinstance(v, dict):
        print("ahead", sys.argv[1], k, {a: b for a, b in v.items() if a not in ("trace", "what", "verdict", "children_cpu_seconds_so_far")}, v.get("verdict", "")[:40])
        if sys.argv[1] == "1":
            for t in v.get("trace", [])[:34]: print("    ", t)
PY
done
