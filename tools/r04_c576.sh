# the check half at 576 slices per frame: how the decoder's rate moves with the batch.  bash tools/r04_c576.sh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
for b in 112 168 224 280 336 392 448 512 576; do
  timeout 600 python bench.py --mode check --steps 2 --warmup 1 --legs "" --slices 576 --batch 112 --check-batch $b > gpurun_out/r04/c576_$b.json 2> gpurun_out/r04/c576_$b.err || tail -3 gpurun_out/r04/c576_$b.err
  python - $b <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r04/c576_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
k = d["roofline"]["kernel_ms"]
print(sys.argv[1], d["value"], "frames/s", d["ms_per_step"], "ms", k, "kernel alone %.1f frames/s" % (d["config"]["frames_per_step_per_gpu"] / k["k_dec_slices"] * 1e3), d["config"]["all_frames_identical_to_source"])
PY
done
