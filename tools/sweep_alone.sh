# Needs the timing build: make -C rawcooked_amd/csrc timing  (librcgpu_timing.so, loaded through RCGPU_LIB).
# k_resolve alone (no range coder launched: timing only) against its occupancy.  Usage on the GPU box: bash tools/sweep_alone.sh "0 12000 13200 16000 19088 24000 40000"
for lds in ${1:-0 12000 13200 16000 19088 24000 40000}; do
  RCGPU_LIB=${RCGPU_LIB:-rawcooked_amd/librcgpu_timing.so} RCGPU_BENCH_TIMING_BUILD=1 RCGPU_EXP_SKIP_RC=1 RCGPU_RESOLVE_LDS_TOTAL=$lds timeout 300 python bench.py --steps 2 --warmup 1 --legs "" --no-verify ${2:-} > /tmp/line.json 2>/tmp/line.err || tail -3 /tmp/line.err
  python3 tools/bench_line.py /tmp/line.json "k_resolve alone, lds $lds"
done
