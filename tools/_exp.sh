export TMPDIR=/tmp
python -m pytest tests/test_seed.py tests/test_mapread.py -m gpu -x -q 2>&1 | tail -2
python bench.py --steps 8 --warmup 3 --no-cpu-baseline --defer-seed 0 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"sketch_count": [0-9.]*\|"sketch_emit": [0-9.]*'
python bench.py --steps 8 --warmup 3 --no-cpu-baseline --defer-seed 0 --lanes 2 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"lane_items": [^]]*]'
python bench.py --steps 8 --warmup 3 --no-cpu-baseline --defer-seed 0 --lanes 2 --lane-priority 0 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"lane_items": [^]]*]'
