export TMPDIR=/tmp
LRA_STAGE_DBG=1 LRA_BENCH_DBG=1 python bench.py --steps 3 --warmup 4 --no-cpu-baseline --heavy-pool 2600 > gpurun_out/r04h_stage.txt 2>&1
grep -n "bench\] lane\|stage\]" gpurun_out/r04h_stage.txt | tail -75
