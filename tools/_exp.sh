export TMPDIR=/tmp
python bench.py --steps 5 --warmup 2 --defer-seed 0 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
LRA_RECORD_THREADS=96 python bench.py --steps 5 --warmup 2 --defer-seed 0 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
LRA_RECORD_THREADS=32 python bench.py --steps 5 --warmup 2 --defer-seed 0 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
LRA_BENCH_DBG=1 python bench.py --steps 4 --warmup 2 --defer-seed 0 --no-cpu-baseline 2>&1 | grep "bench\]" | tail -6
