# bench.py over hand-over segment counts: bash tools/sweep_segments.sh 16 24 32
for sg in "$@"; do
  python bench.py --steps 2 --legs "" --no-verify --segments $sg 2>&1 | tail -1 | SG=$sg python -c "import sys,json,os; d=json.loads(sys.stdin.read()); print(os.environ['SG'], d['value'], d['ms_per_step'], d['config']['hbm_in_use_gb'], d['roofline']['kernel_ms_per_step'])"
done
