# Needs the timing build: make -C rawcooked_amd/csrc timing  (the shipped library reads none of these variables).
# k_resolve occupancy sweep: forced workgroup LDS totals x range-coder mappings.
# Usage on the GPU box: bash tools/sweep_occ.sh "0 12000 13200 16000 19088" "64 1" [extra bench args]
for sp in ${2:-64 1}; do for lds in ${1:-0 12000 13200 16000 19088}; do
  RCGPU_LIB=${RCGPU_LIB:-rawcooked_amd/librcgpu_timing.so} RCGPU_RC_SPAN=$sp RCGPU_RESOLVE_LDS_TOTAL=$lds timeout 300 python bench.py --steps 2 --warmup 1 --legs "" --no-verify ${3:-} > /tmp/line.json 2>/tmp/line.err || tail -3 /tmp/line.err
  python3 tools/bench_line.py /tmp/line.json "span $sp lds $lds"
done; done
