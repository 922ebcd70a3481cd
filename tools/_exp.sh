export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 4 --warmup 2 --defer-seed 0 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' > gpurun_out/r04n_buildAll.json
LRA_STAGE_DBG=1 python bench.py --steps 2 --warmup 1 --defer-seed 0 --no-cpu-baseline 2>&1 | grep "stage\] sdp" | tail -4
for f in gpurun_out/r04n_*.json; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f) $(grep -o '"sdp_build": [0-9.]*' $f) $(grep -o '"sdp_process": [0-9.]*' $f) $(grep -o '"sdp_inner_build": [0-9.]*' $f); done
