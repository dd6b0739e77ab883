set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_check.py tests/test_gpu_pipeline.py -x -q -m gpu > gpurun_out/r04/t1.log 2>&1; tail -3 gpurun_out/r04/t1.log
timeout 600 python bench.py --mode check --steps 2 --warmup 1 --legs "" > gpurun_out/r04/check_a.json 2> gpurun_out/r04/check_a.err; tail -c 1500 gpurun_out/r04/check_a.json
timeout 600 python bench.py --mode check --steps 2 --warmup 1 --legs "" --slices 576 --check-batch 256 > gpurun_out/r04/check_576.json 2> gpurun_out/r04/check_576.err; tail -c 600 gpurun_out/r04/check_576.json
PASS_TIMEOUT=500 bash tools/profile_check.sh r04a > gpurun_out/r04/prof_check.log 2>&1; tail -5 gpurun_out/r04/prof_check.log
grep k_dec_slices gpurun_out/summary/r04a_check_pmc_*.csv
