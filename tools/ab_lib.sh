# A/B of library builds under one environment.  Usage on the GPU box: bash tools/ab_lib.sh "<env settings>" lib1.so lib2.so ...   ("product" = the built library)
cp rawcooked_amd/librcgpu.so /tmp/librcgpu_product.so
E=$1; shift
for lib in "$@"; do
  if [ "$lib" = product ]; then cp /tmp/librcgpu_product.so rawcooked_amd/librcgpu.so; else cp $lib rawcooked_amd/librcgpu.so; fi
  env $E RCGPU_BENCH_TIMING_BUILD=1 timeout 300 python bench.py --steps 2 --warmup 1 --legs "" --no-verify > /tmp/line.json 2>/tmp/line.err || tail -3 /tmp/line.err
  python3 tools/bench_line.py /tmp/line.json "$lib [$E]"
done
cp /tmp/librcgpu_product.so rawcooked_amd/librcgpu.so
