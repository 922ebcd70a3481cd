#!/bin/bash
# Same-box A/B of two builds of the library under rocprofv3 --kernel-trace: ab_base/liblra_hip.so (A) against lra_amd/liblra_hip.so (B), the one call per step
# (nothing beside it), the rows of the kernels matching <pattern>.  usage (on the GPU box): bash tools/ab_trace.sh <tag> <pattern> [bench flags]
export TMPDIR=/tmp
tag=$1; pat=$2; shift 2
cp lra_amd/liblra_hip.so /tmp/ab_new.so
for v in A B; do
  if [ $v = A ]; then cp ab_base/liblra_hip.so lra_amd/liblra_hip.so; else cp /tmp/ab_new.so lra_amd/liblra_hip.so; fi
  rm -rf /tmp/kt_${tag}_$v; mkdir -p /tmp/kt_${tag}_$v
  rocprofv3 --kernel-trace --stats -d /tmp/kt_${tag}_$v -o p -- python bench.py --steps 3 --warmup 1 --two-stage 0 --seed-ahead 0 --no-records --no-cpu-baseline "$@" > /tmp/kt_${tag}_$v/log.txt 2>&1
  python tools/rocpd_summary.py $(ls /tmp/kt_${tag}_$v/*.db | head -1) > gpurun_out/${tag}_${v}_ks.txt
  echo "== $v $(grep -o '"ms_per_step": [0-9.]*' /tmp/kt_${tag}_$v/log.txt)"
  grep -E "$pat" gpurun_out/${tag}_${v}_ks.txt | cut -c1-60,111-190
done
cp /tmp/ab_new.so lra_amd/liblra_hip.so
