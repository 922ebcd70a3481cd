#!/bin/bash
# rocprofv3 --kernel-trace of tools/bench_presets.py --preset <p>; the per-kernel table goes to gpurun_out/<tag>_<p>_kernel_stats.txt.  usage: bash tools/trace_preset.sh <tag> <preset> [flags]
export TMPDIR=/tmp
tag=$1; p=$2; shift 2
rm -rf /tmp/kp_${tag}_$p; mkdir -p /tmp/kp_${tag}_$p
rocprofv3 --kernel-trace --stats -d /tmp/kp_${tag}_$p -o p -- python tools/bench_presets.py --preset $p "$@" > /tmp/kp_${tag}_$p/log.txt 2>&1
python tools/rocpd_summary.py $(ls /tmp/kp_${tag}_$p/*.db | head -1) > gpurun_out/${tag}_${p}_kernel_stats.txt
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' /tmp/kp_${tag}_$p/log.txt | tr '\n' ' '; echo
head -16 gpurun_out/${tag}_${p}_kernel_stats.txt | cut -c1-64,111-175
