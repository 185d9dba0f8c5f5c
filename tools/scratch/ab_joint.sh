mkdir -p gpurun_out/jb
L=$PWD/ts-asr_amd/lib
for i in 1 2; do
  for v in base kb2o2 kb2o1; do
    if [ $v = base ]; then unset TSASR_HIP_LIB; else export TSASR_HIP_LIB=$L/libtsasr_hip_$v.so; fi
    python bench.py --no-cpu-baseline > gpurun_out/jb/h_${v}_$i.json 2> gpurun_out/jb/h_${v}_$i.err || exit 1
    python bench.py --config longform --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/jb/l_${v}_$i.json 2> gpurun_out/jb/l_${v}_$i.err || exit 1
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/jb/*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], d['ms_per_step'], d.get('rnnt_joint_loss_ms'))
PY
