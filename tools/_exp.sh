export TMPDIR=/tmp
bash tools/profile_bench.sh r04n --steps 20 --warmup 5 > /dev/null 2>&1
python -c "
import json
j=json.loads(open('gpurun_out/r04n_bench.json').read()); print(j['value'], j['ms_per_step'], j['cpu_baseline']['value'], j['cpu_baseline']['sample_equals_gpu'], j['roofline']['frac'], j['roofline']['avg_launch_ms'], j['hbm_used_gb'])"
for p in clr ccs; do python tools/bench_presets.py --preset $p --steps 5 2>/dev/null | grep '^{"metric"' > gpurun_out/r04n_${p}_bench.json; echo $p $(grep -o '"value": [0-9.]*' gpurun_out/r04n_${p}_bench.json); done
