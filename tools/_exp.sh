export TMPDIR=/tmp
python -m pytest tests/test_sdp.py tests/test_mapread.py tests/test_highacc_path.py tests/test_local.py tests/test_refine_splitchain.py -m gpu -x -q 2>&1 | tail -3
for v in "LRA_SDP_TWO=1" "LRA_SDP_TWO=0"; do
echo "$v: $(env $v LRA_STAGE_DBG=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-records 2>&1 | grep 'stage\] sdp#A\|stage\] sdp#2\|ms_per_step\|"sdp_process":' | tail -3 | sed 's/.*sdp#A *//; s/.*sdp#2 *//; s/.*"ms_per_step": \([0-9.]*\).*"sdp_process": \([0-9.]*\).*/step \1 proc \2/' | tr '\n' ' ')"
done
