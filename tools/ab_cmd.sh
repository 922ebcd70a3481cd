#!/bin/bash
# Same-box A/B of two builds of the library on any command that prints bench-style JSON lines: ab_base/liblra_hip.so (A), lra_amd/liblra_hip.so (B), alternating.
# usage (on the GPU box): bash tools/ab_cmd.sh <pairs> <command...>
pairs=$1; shift
cp lra_amd/liblra_hip.so /tmp/ab_new.so
for i in $(seq 1 $pairs); do
  for v in A B; do
    if [ $v = A ]; then cp ab_base/liblra_hip.so lra_amd/liblra_hip.so; else cp /tmp/ab_new.so lra_amd/liblra_hip.so; fi
    "$@" 2>/dev/null | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$v$i', round(d['value'],4), round(d['ms_per_step'],1))"
  done
done
cp /tmp/ab_new.so lra_amd/liblra_hip.so
