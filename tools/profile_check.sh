# rocprofv3 of the check half (config 5): --stats, then the two PMC passes that give k_dec_slices its HBM bytes.  GPU box: bash tools/profile_check.sh <tag>
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
# (until round 5 this script set RCGPU_NO_CU_PARTITION=1: rocprofv3 lost its trace at exit in a process that still held a CU-masked stream.  bench.py now
# gives the library's hash streams back before it ends -- rcgpu_release_device_streams, tools/rocprof_cumask_repro.py -- and the profile is of the configuration the line times)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/pc gpurun_out/summary
timeout ${PASS_TIMEOUT:-420} rocprofv3 --kernel-trace --stats -d gpurun_out/pc -o chk -- python bench.py --mode check --steps 2 --warmup 1 --legs "" > gpurun_out/pc/log 2>&1
python tools/rocprof_summary.py stats $(find gpurun_out/pc -name "*.db" | head -1) > gpurun_out/summary/${TAG}_check_kernel_stats.csv
grep -o '{"metric.*' gpurun_out/pc/log | head -1 > gpurun_out/summary/${TAG}_check_bench_under_rocprof.json
timeout ${PASS_TIMEOUT:-420} rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pc/f -o chk -- python bench.py --mode check --steps 1 --warmup 0 --legs "" > gpurun_out/pc/logf 2>&1
timeout ${PASS_TIMEOUT:-420} rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pc/w -o chk -- python bench.py --mode check --steps 1 --warmup 0 --legs "" > gpurun_out/pc/logw 2>&1
python tools/rocprof_summary.py pmc $(find gpurun_out/pc/f -name "*.db" | head -1) FETCH_SIZE > gpurun_out/summary/${TAG}_check_pmc_fetch_size.csv
python tools/rocprof_summary.py pmc $(find gpurun_out/pc/w -name "*.db" | head -1) WRITE_SIZE > gpurun_out/summary/${TAG}_check_pmc_write_size.csv
rm -rf gpurun_out/pc
cat gpurun_out/summary/${TAG}_check_kernel_stats.csv; cut -c1-300 gpurun_out/summary/${TAG}_check_bench_under_rocprof.json
