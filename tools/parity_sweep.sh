#!/bin/bash
# The headline bench's oracle comparison (cpu_baseline.sample_equals_gpu) over a few workload variants, N reads of each through the oracle.
# usage (on the GPU box): bash tools/parity_sweep.sh <N> "<flags a>" "<flags b>" ...
N=$1; shift
for f in "$@"; do
  LRA_BENCH_CPU_SAMPLE=$N python bench.py --steps 1 --warmup 1 $f 2>/dev/null | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); c=d['cpu_baseline']; print('[$f]', 'equal' if c['sample_equals_gpu'] else 'DIFFERENT', c['sample'][:70], 'flagged', d['config']['per_step'].get('n_flagged_reads'))"
done
