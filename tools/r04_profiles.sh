# round-4 profile set: encode (r04b) and check (r04b_check), then the alias run of the N>1 bench record.  bash tools/r04_profiles.sh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04 gpurun_out/summary
bash tools/profile_r03.sh r04b > gpurun_out/r04/prof_enc.log 2>&1; tail -30 gpurun_out/r04/prof_enc.log | cut -c1-200
cp gpurun_out/prof/r04b_* gpurun_out/summary/ 2>/dev/null
PASS_TIMEOUT=500 bash tools/profile_check.sh r04b > gpurun_out/r04/prof_chk.log 2>&1; tail -6 gpurun_out/r04/prof_chk.log | cut -c1-200
timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --legs host --host-frames 1344 > gpurun_out/r04/bench_alias2.json 2> gpurun_out/r04/bench_alias2.err; tail -2 gpurun_out/r04/bench_alias2.err | cut -c1-300
python tools/show_line.py gpurun_out/r04/bench_alias2.json | grep -n "single_process" -A40 | cut -c1-220
