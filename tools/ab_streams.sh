for n in 2 3 4 2 3; do
  PM_ENCODER_STREAMS=$n python bench.py --workload vit --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/st_$n.json 2>/dev/null || exit 1
  python -c "
import json; d=json.loads(open('gpurun_out/st_$n.json').read().strip().splitlines()[-1]); print('streams=$n', d['value'], d['ms_per_step'])"
done
