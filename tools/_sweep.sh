#!/bin/bash
# usage: sweep.sh tag "ENV=.. ENV2=.." [bench args]
tag=$1; envs=$2; shift 2
env $envs python bench.py --steps 8 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; a=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', round(a['ms_per_step'],1), round(a['value'],4), a['hbm_used_gb'], a['seed_ahead']['on'])" >> gpurun_out/r06o_sweep.txt
