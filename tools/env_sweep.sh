#!/bin/bash
# bench.py under a list of environment settings on one box: usage: bash tools/env_sweep.sh "<bench flags>" "VAR=a" "VAR=b" ... ("-" = none)
flags=$1; shift
for e in "$@"; do
  if [ "$e" = "-" ]; then env_cmd=""; else env_cmd="env $e"; fi
  $env_cmd python bench.py $flags 2>/dev/null | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); k=d['kernel_ms_per_step']; print('$e', round(d['value'],4), round(d['ms_per_step'],1), {x: round(k[x],1) for x in ('sdp_process','sdp_process_wg','sdp_build') if x in k})"
done
