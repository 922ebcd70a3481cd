export TMPDIR=/tmp
python -m pytest tests/test_sdp.py -m gpu -x -q 2>&1 | tail -2
for v in "X=1" "LRA_SDP_BUILD16_TWO=0"; do
echo "$v: $(env $v LRA_STAGE_DBG=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-records 2>&1 | grep 'stage\] sdp#2\|ms_per_step' | tail -2 | sed 's/.*sdp#2 *//; s/.*"ms_per_step": \([0-9.]*\).*/step \1/' | tr '\n' ' ')"
done
mkdir -p /tmp/q1; rocprofv3 --kernel-trace --stats -d /tmp/q1 -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-records > /tmp/q1/log.txt 2>&1
DB=$(ls /tmp/q1/*.db | head -1); python tools/trace_export.py $DB 1.4 2000 | grep -E "sdp_build|sdp_process" | awk -F'\t' '{printf "%9.1f %7.1f s%s g%s %s\n",$1,$2,$3,$5,substr($NF,1,45)}' | tail -8
