export TMPDIR=/tmp
python -m pytest tests/test_refine.py tests/test_mapread.py tests/test_highacc_path.py tests/test_highacc.py -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 5 --warmup 2 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"ir_band": [0-9.]*\|"sample_equals_gpu": [a-z]*'
python tools/bench_presets.py --preset clr --steps 4 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*'
