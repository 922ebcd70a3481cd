#!/bin/bash
# bench.py under a list of flag sets on one box: usage: bash tools/flag_sweep.sh "<common flags>" "<flags a>" "<flags b>" ...
common=$1; shift
for f in "$@"; do
  python bench.py $common $f 2>/dev/null | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('[$f]', round(d['value'],4), round(d['ms_per_step'],1))"
done
