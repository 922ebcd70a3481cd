#!/bin/bash
# usage (on the GPU box): tools/profile_counters.sh <tag> "<COUNTER1 COUNTER2 ...>" [bench args...]
# One rocprofv3 --pmc pass (kernel-trace only) over bench.py; per-kernel sums of every counter -> gpurun_out/<tag>_pmc_<COUNTER>.txt
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
ctrs=$1; shift
mkdir -p $R/gpurun_out
cd $R
rm -rf /tmp/pmcx; mkdir -p /tmp/pmcx
rocprofv3 --pmc $ctrs --kernel-trace -d /tmp/pmcx -o p -- python bench.py "$@" --no-cpu-baseline > /tmp/pmcx/log.txt 2>&1
tail -2 /tmp/pmcx/log.txt | cut -c1-200
for C in $ctrs; do
  python tools/rocpd_pmc.py $(ls /tmp/pmcx/*.db | head -1) $C > $R/gpurun_out/${tag}_pmc_$C.txt 2>&1
  echo "== $C"; grep -E "sdp_|kernel" $R/gpurun_out/${tag}_pmc_$C.txt | head -8
done
