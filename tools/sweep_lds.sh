# Timing sweep (wrong bytes): tools/bin/librcgpu_fold.so (k_resolve's collision table folded to 1 KB) at several workgroup LDS sizes, with
# the whole-slice and the split range coder.  Usage on the GPU box: bash tools/sweep_lds.sh "9600 10600 13000 16000 19088" "64 1"
cp rawcooked_amd/librcgpu.so /tmp/librcgpu_keep.so
cp tools/bin/librcgpu_fold.so rawcooked_amd/librcgpu.so
for sp in ${2:-64 1}; do for lds in ${1:-9600 10600 13000 16000 19088}; do
  RCGPU_BENCH_TIMING_BUILD=1 RCGPU_RC_SPAN=$sp RCGPU_RESOLVE_LDS_TOTAL=$lds timeout 300 python bench.py --steps 2 --warmup 1 --legs "" --no-verify ${3:-} > /tmp/line.json 2>/tmp/line.err || tail -3 /tmp/line.err
  python3 tools/bench_line.py /tmp/line.json "span $sp lds $lds"
done; done
cp /tmp/librcgpu_keep.so rawcooked_amd/librcgpu.so
