#!/bin/bash
# usage (on the GPU box): tools/kernel_trace.sh <tag>
# rocprofv3 --kernel-trace of bench.py --steps 2 --warmup 1 --no-records; the per-kernel table goes to gpurun_out/<tag>_ks.txt.  For A / B runs of two
# builds of the library on ONE box (box-to-box spread is +-3 %): keep the other build beside the tree (tools/_base.so, git-ignored), copy it over
# lra_amd/liblra_hip.so between two calls.
export TMPDIR=/tmp
tag=$1
rm -rf /tmp/kt_$tag; mkdir -p /tmp/kt_$tag
rocprofv3 --kernel-trace --stats -d /tmp/kt_$tag -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-records > /tmp/kt_$tag/log.txt 2>&1
python tools/rocpd_summary.py $(ls /tmp/kt_$tag/*.db | head -1) > gpurun_out/${tag}_ks.txt
grep -o '"ms_per_step": [0-9.]*' /tmp/kt_$tag/log.txt
