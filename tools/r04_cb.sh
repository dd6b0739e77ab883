cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
for cb in 1024 1344 2016; do
timeout 600 python bench.py --mode check --steps 2 --warmup 1 --legs "" --check-batch $cb > gpurun_out/r04/check_cb$cb.json 2> gpurun_out/r04/check_cb$cb.err || tail -3 gpurun_out/r04/check_cb$cb.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r04/check_cb$cb.json").read().strip().splitlines()[-1])
print($cb, d["value"], "frames/s", d["ms_per_step"], "ms", d["roofline"]["kernel_ms"], d["config"]["all_frames_identical_to_source"])
PY
done
