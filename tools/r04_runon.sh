# run-on mode A/B: the encoder's tests, then the device-resident steps with and without it.  bash tools/r04_runon.sh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 1200 python -m pytest tests/test_gpu_stages.py tests/test_gpu_configs.py tests/test_gpu_pipeline.py -x -q -m gpu 2>&1 | tail -3
for ro in 1 0 1 0; do
  timeout 600 python bench.py --steps 6 --warmup 2 --legs "" --run-on $ro > gpurun_out/r04/ro_$ro.json 2> gpurun_out/r04/ro_$ro.err || tail -5 gpurun_out/r04/ro_$ro.err
  python - $ro <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r04/ro_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("run-on", sys.argv[1], d["value"], "frames/s", d["ms_per_step"], "ms", d["roofline"]["kernel_ms_per_step"], d["config"]["verified_vs_oracle"], d["config"]["verified_by_reference"], d["config"]["device_error_flags"], d["config"]["hbm_in_use_gb"])
PY
done
