# interleaved A/B of the ViT leg: $1 = env assignment for variant B (e.g. PM_ATTN_SHORT=0), prints img/s and per-kernel us
for i in 1 2 3; do
  python bench.py --workload vit --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab_a$i.json 2>/dev/null || exit 1
  env $1 python bench.py --workload vit --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab_b$i.json 2>/dev/null || exit 1
done
python - <<'PY'
import json
for f in sorted(__import__('glob').glob('gpurun_out/ab_[ab]?.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], {k: round(v['total_ms'] / v['launches'] * 1e3, 1) for k, v in d['kernels'].items()})
PY
