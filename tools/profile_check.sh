cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/pc gpurun_out/summary
rocprofv3 --kernel-trace --stats -d gpurun_out/pc -o chk -- python bench.py --mode check --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pc/log 2>&1
python tools/rocprof_summary.py stats $(find gpurun_out/pc -name "*.db" | head -1) > gpurun_out/summary/r01e_check_kernel_stats.csv
grep -o '{"metric.*' gpurun_out/pc/log | head -1 > gpurun_out/summary/r01e_check_bench_under_rocprof.json
rm -rf gpurun_out/pc
cat gpurun_out/summary/r01e_check_kernel_stats.csv; cut -c1-300 gpurun_out/summary/r01e_check_bench_under_rocprof.json
