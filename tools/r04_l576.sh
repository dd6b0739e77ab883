# the linked reference's --check at 576 slices per frame, frames per device call as RCGPU_LINKED_BATCH says.  bash tools/r04_l576.sh "256 384 512"
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
for b in $1; do
  if [ "$b" = auto ]; then unset RCGPU_LINKED_BATCH; else export RCGPU_LINKED_BATCH=$b; fi
  RCGPU_TRACE_KEPT=1 timeout 900 python bench.py --mode check --legs cpu --steps 1 --warmup 0 --slices 576 --check-batch 336 > gpurun_out/r04/l576_$b.json 2> gpurun_out/r04/l576_$b.err; tail -2 gpurun_out/r04/l576_$b.err | cut -c1-200
  python - $b <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r04/l576_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
l = d["linked_check"]
for k, v in l.items():
    if isinstance(v, dict):
        print(sys.argv[1], k, {a: b for a, b in v.items() if a not in ("trace", "what", "verdict")}, v.get("verdict", "")[:40])
        for t in v.get("trace", [])[:40]: print("    ", t)
PY
done
