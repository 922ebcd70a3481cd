export TMPDIR=/tmp
bash tools/profile_bench.sh r04m --steps 10 --warmup 3 > /dev/null 2>&1
bash tools/profile_pmc.sh r04m --steps 3 --warmup 1 > /dev/null 2>&1
LRA_BENCH_CPU_SAMPLE=32768 python bench.py --steps 3 --warmup 1 2>/dev/null | grep '^{"metric"' > gpurun_out/r04m_full_batch_parity.json
head -c 400 gpurun_out/r04m_bench.json; echo; head -12 gpurun_out/r04m_kernel_stats.txt | cut -c1-170; grep -o '"cpu_baseline": {[^}]*}' gpurun_out/r04m_full_batch_parity.json | cut -c1-300
