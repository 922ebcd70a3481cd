export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 2>/dev/null | grep '^{"metric"' > gpurun_out/r04ai_bench.json
python - <<'P'
import json
j=json.loads(open('gpurun_out/r04ai_bench.json').read())
print(j['value'], j['ms_per_step'], j['cpu_baseline']['value'], j['cpu_baseline']['cores'], j['cpu_baseline'].get('sample_equals_gpu'), j['hbm_used_gb'])
P
