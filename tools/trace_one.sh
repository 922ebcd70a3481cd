#!/bin/bash
# rocprofv3 --kernel-trace of the one call per step with the library as it is; rows matching <pattern>.  usage: [ENV=..] bash tools/trace_one.sh <tag> <pattern> [bench flags]
export TMPDIR=/tmp
tag=$1; pat=$2; shift 2
rm -rf /tmp/kt_${tag}; mkdir -p /tmp/kt_${tag}
rocprofv3 --kernel-trace --stats -d /tmp/kt_${tag} -o p -- python bench.py --steps 3 --warmup 1 --two-stage 0 --seed-ahead 0 --no-records --no-cpu-baseline "$@" > /tmp/kt_${tag}/log.txt 2>&1
python tools/rocpd_summary.py $(ls /tmp/kt_${tag}/*.db | head -1) > gpurun_out/${tag}_ks.txt
echo "== $tag $(grep -o '"ms_per_step": [0-9.]*' /tmp/kt_${tag}/log.txt)"
grep -E "$pat" gpurun_out/${tag}_ks.txt | cut -c1-60,111-190
