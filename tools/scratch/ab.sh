set -e
for cfg in "TSASR_EARLY_FLUSH=0" "TSASR_EARLY_FLUSH=1" "TSASR_EARLY_FLUSH=0" "TSASR_EARLY_FLUSH=1"; do
  echo "== $cfg"
  env $cfg timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['loss'])"
done
