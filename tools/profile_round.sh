set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R
rocprofv3 --kernel-trace --stats -d gpurun_out/prof/stats -o r01c -- python bench.py --steps 3 --warmup 1 --legs "" --no-verify > gpurun_out/prof/bench_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof/pmc_fetch -o r01c -- python bench.py --steps 1 --warmup 0 --legs "" --no-verify > gpurun_out/prof/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof/pmc_write -o r01c -- python bench.py --steps 1 --warmup 0 --legs "" --no-verify > gpurun_out/prof/bench_write.log 2>&1
find gpurun_out/prof -type f | head -50
