# decoder A/B on the GPU box: the check tests, then the check half at 64 and 576 slices.  bash tools/r04_check_ab.sh <tag>
TAG=${1:-x}; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_check.py tests/test_gpu_stages.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py --mode check --steps 2 --warmup 1 --legs "" > gpurun_out/r04/check_$TAG.json 2> gpurun_out/r04/check_$TAG.err || tail -3 gpurun_out/r04/check_$TAG.err
timeout 600 python bench.py --mode check --steps 2 --warmup 1 --legs "" --slices 576 --check-batch 256 > gpurun_out/r04/check576_$TAG.json 2> gpurun_out/r04/check576_$TAG.err || tail -3 gpurun_out/r04/check576_$TAG.err
python - <<PY
import json
for f in ["check_$TAG", "check576_$TAG"]:
    d = json.loads(open("gpurun_out/r04/%s.json" % f).read().strip().splitlines()[-1])
    print(f, d["value"], "frames/s", d["ms_per_step"], "ms", d["config"]["frames_per_step_per_gpu"], d["roofline"]["kernel_ms"], d["config"]["all_frames_identical_to_source"])
PY
