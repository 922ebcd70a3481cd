# usage: tools/_kt.sh <tag> : kernel-trace of 2 steps, per-kernel ms per step for the sdp kernels
export TMPDIR=/tmp
tag=$1
rm -rf /tmp/kt_$tag; mkdir -p /tmp/kt_$tag
rocprofv3 --kernel-trace --stats -d /tmp/kt_$tag -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-records > /tmp/kt_$tag/log.txt 2>&1
python tools/rocpd_summary.py $(ls /tmp/kt_$tag/*.db | head -1) > gpurun_out/${tag}_ks.txt
grep -o '"ms_per_step": [0-9.]*' /tmp/kt_$tag/log.txt
