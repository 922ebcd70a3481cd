#!/bin/bash
# usage (on the GPU box): tools/profile_pmc.sh <tag> [bench args...]
# Two separate counter passes (FETCH_SIZE, WRITE_SIZE) with --kernel-trace only, as the MI355X guide
# prescribes; keeps only per-kernel text summaries under gpurun_out/.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
mkdir -p $R/gpurun_out
cd $R
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C; mkdir -p /tmp/pmc_$C
  rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_$C -o p -- python bench.py "$@" --no-cpu-baseline > /tmp/pmc_$C/log.txt 2>&1
  python tools/rocpd_pmc.py $(ls /tmp/pmc_$C/*.db | head -1) $C > $R/gpurun_out/${tag}_pmc_$C.txt 2>&1
  tail -3 /tmp/pmc_$C/log.txt | cut -c1-300
done
head -40 $R/gpurun_out/${tag}_pmc_FETCH_SIZE.txt
