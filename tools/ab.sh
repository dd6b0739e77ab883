# A/B two builds of librcgpu.so in one session: bash tools/ab.sh <old.so> <new.so> [rounds]
for r in $(seq 1 ${3:-2}); do
  for v in $1 $2; do
    cp $v rawcooked_amd/librcgpu.so
    python bench.py --steps 2 --legs "" --no-verify 2>&1 | tail -1 | V=$v python -c "import sys,json,os; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']; print(os.environ['V'], d['value'], d['ms_per_step'], k['k_resolve'], k['k_rangecode'])"
  done
done
