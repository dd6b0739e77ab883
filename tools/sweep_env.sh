# A/B over environment settings with the timing build (make -C rawcooked_amd/csrc timing): the shipped library reads no measuring switch.  Usage on the GPU box: bash tools/sweep_env.sh "A=1 B=2" "A=3" ...   (each argument one run)
for cfg in "$@"; do
  env RCGPU_LIB=${RCGPU_LIB:-rawcooked_amd/librcgpu_timing.so} $cfg timeout 300 python bench.py --steps 2 --warmup 1 --legs "" --no-verify > /tmp/line.json 2>/tmp/line.err || tail -3 /tmp/line.err
  python3 tools/bench_line.py /tmp/line.json "$cfg"
done
