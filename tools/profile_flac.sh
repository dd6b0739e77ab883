# rocprofv3 --stats of the FLAC kernels on config 3's audio.  GPU box: bash tools/profile_flac.sh <tag>
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/pf gpurun_out/summary
rocprofv3 --kernel-trace --stats -d gpurun_out/pf -o flac -- python tools/flac_bench.py > gpurun_out/pf/log 2>&1
python tools/rocprof_summary.py stats $(find gpurun_out/pf -name "*.db" | head -1) > gpurun_out/summary/${TAG}_flac_kernel_stats.csv
grep samples_per_channel gpurun_out/pf/log > gpurun_out/summary/${TAG}_flac_bench.json
rm -rf gpurun_out/pf
cat gpurun_out/summary/${TAG}_flac_kernel_stats.csv gpurun_out/summary/${TAG}_flac_bench.json
