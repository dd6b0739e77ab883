# the driver's bench line, pretty-printed.  bash tools/r04_bench.sh <tag> [bench args]
TAG=${1:-r04}; shift; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
T0=$(date +%s)
timeout 1700 python bench.py "$@" > gpurun_out/r04/bench_$TAG.json 2> gpurun_out/r04/bench_$TAG.err; echo "rc $? wall $(( $(date +%s) - T0 )) s"; tail -3 gpurun_out/r04/bench_$TAG.err | cut -c1-300
python tools/show_line.py gpurun_out/r04/bench_$TAG.json > gpurun_out/r04/bench_$TAG.txt 2>&1; grep -n "value\|seconds\|frac\|error\|verdict\|floor" gpurun_out/r04/bench_$TAG.txt | head -120
