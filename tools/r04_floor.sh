# The two numbers the round-3 verdict asked for beside k_resolve's byte roofline (GPU box: bash tools/r04_floor.sh <tag>):
#   (a) tools/bin/gather_peak: the chip's ceiling in random 32-byte records gathered and written back;
#   (b) k_resolve with every state record in an L2-resident region (timing build, RCGPU_EXP_STATES_L2: right work, wrong bytes): its floor
#       without the HBM latency of the state traffic -- alone and beside the range coder.
TAG=${1:-r04}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$TAG
timeout 600 tools/bin/gather_peak 7 > gpurun_out/$TAG/gather_peak.jsonl 2> gpurun_out/$TAG/gather_peak.err; tail -3 gpurun_out/$TAG/gather_peak.jsonl
export RCGPU_LIB=rawcooked_amd/librcgpu_timing.so RCGPU_BENCH_TIMING_BUILD=1
run() {  # label, env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps 3 --warmup 1 --legs "" --no-verify > /tmp/line.json 2>/tmp/line.err || tail -3 /tmp/line.err
  python3 tools/bench_line.py /tmp/line.json "$label" | tee -a gpurun_out/$TAG/floor.txt
}
run "product mapping (timing build, nothing set)" X=1
run "k_resolve alone (no range coder)" RCGPU_EXP_SKIP_RC=1
run "k_resolve alone, states L2-resident" RCGPU_EXP_SKIP_RC=1 RCGPU_EXP_STATES_L2=1
run "states L2-resident, beside the whole-slice coder" RCGPU_EXP_STATES_L2=1
run "split coder" RCGPU_RC_SPAN=64
run "split coder, states L2-resident" RCGPU_RC_SPAN=64 RCGPU_EXP_STATES_L2=1
