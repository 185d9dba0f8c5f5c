# run-to-run reproducibility of the default bench loss (graph + forked branch; no host synchronisation inside the timed region)
for i in 1 2 3 4 5 6; do
  timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(repr(d['loss']), d['ms_per_step'], end=' | ')"
done; echo
