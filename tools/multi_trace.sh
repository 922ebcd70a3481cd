#!/bin/bash
# Same-box comparison of several builds of the library under rocprofv3 --kernel-trace (the one call per step): the tree's own, then every ab_var/*.so.
# usage (on the GPU box): bash tools/multi_trace.sh <tag> <pattern> [bench flags]
tag=$1; pat=$2; shift 2
cp lra_amd/liblra_hip.so /tmp/mt_new.so
bash tools/trace_one.sh ${tag}_tree "$pat" "$@"
for f in ab_var/*.so; do
  n=$(basename $f .so); cp $f lra_amd/liblra_hip.so
  bash tools/trace_one.sh ${tag}_$n "$pat" "$@"
done
cp /tmp/mt_new.so lra_amd/liblra_hip.so
