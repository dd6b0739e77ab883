cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
RCGPU_TRACE_KEPT=1 timeout 900 python bench.py --mode check --legs cpu --steps 1 --warmup 0 > gpurun_out/r04/linked_$1.json 2> gpurun_out/r04/linked_$1.err; tail -2 gpurun_out/r04/linked_$1.err | cut -c1-200
python - <<PY
import json
d = json.loads(open("gpurun_out/r04/linked_$1.json").read().strip().splitlines()[-1])
l = d["linked_check"]
for k, v in l.items():
    if isinstance(v, dict):
        print(k, {a: b for a, b in v.items() if a not in ("trace", "what", "verdict")}, v.get("verdict", "")[:40])
        for t in v.get("trace", [])[:60]: print("    ", t)
    elif k != "what": print(k, v)
for t in l.get("whole_product_trace", [])[:80]: print("   wp:", t)
PY
