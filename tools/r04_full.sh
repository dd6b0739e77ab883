# full GPU suite + the driver's bench line.  bash tools/r04_full.sh <tag>
TAG=${1:-r04}; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r04/gputest_$TAG.log 2>&1; tail -4 gpurun_out/r04/gputest_$TAG.log
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench_$TAG.json 2> gpurun_out/r04/bench_$TAG.err; tail -3 gpurun_out/r04/bench_$TAG.err | cut -c1-300
python tools/show_line.py gpurun_out/r04/bench_$TAG.json > gpurun_out/r04/bench_$TAG.txt 2>&1; grep -n "value\|seconds\|frac\|error\|verdict" gpurun_out/r04/bench_$TAG.txt | head -90
