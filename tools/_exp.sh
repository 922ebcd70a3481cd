export TMPDIR=/tmp
python -m pytest tests/test_sort.py tests/test_sdp.py tests/test_mapread.py tests/test_highacc_path.py -m gpu -x -q 2>&1 | tail -2
python bench.py --steps 8 --warmup 2 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"sort": [0-9.]*\|"sdp_sort": [0-9.]*\|"sdp_inner_sort": [0-9.]*\|"sample_equals_gpu": [a-z]*'
