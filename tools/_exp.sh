export TMPDIR=/tmp
for p in ccs clr contig; do python tools/bench_presets.py --preset $p --steps 5 2>/dev/null | grep '^{"metric"' > gpurun_out/r04m_${p}_bench.json; echo $p $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"reads_flagged": [0-9]*\|"reads_with_an_alignment": [0-9]*' gpurun_out/r04m_${p}_bench.json | tr '\n' ' '); done
echo lowprio $(python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
sed -i 's|, priority=prio_lo)   # (the copy|)   # (the copy|' bench.py
echo normal $(python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
