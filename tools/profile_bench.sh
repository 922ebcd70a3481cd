#!/bin/bash
# usage (on the GPU box, via gpurun): tools/profile_bench.sh <tag> [bench args...]
# Runs bench.py once plain (JSON line -> gpurun_out/<tag>_bench.json) and once under
# rocprofv3 --kernel-trace --stats; keeps only the text summary (-> gpurun_out/<tag>_kernel_stats.txt).
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
mkdir -p $R/gpurun_out /tmp/prof_$tag
cd $R
python bench.py "$@" 2>/dev/null | grep '^{"metric"' > $R/gpurun_out/${tag}_bench.json
rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o p -- python bench.py "$@" --no-cpu-baseline > /tmp/prof_$tag/log.txt 2>&1
python tools/rocpd_summary.py $(ls /tmp/prof_$tag/*.db | head -1) > $R/gpurun_out/${tag}_kernel_stats.txt
grep '^{"metric"' /tmp/prof_$tag/log.txt > $R/gpurun_out/${tag}_bench_under_rocprof.json
head -c 1500 $R/gpurun_out/${tag}_bench.json; echo
