export TMPDIR=/tmp
cp lra_amd/liblra_hip.so /tmp/orig.so
for v in orig occ3; do
  if [ $v = orig ]; then cp /tmp/orig.so lra_amd/liblra_hip.so; else cp _var/liblra_$v.so lra_amd/liblra_hip.so; fi
  echo "$v: $(LRA_STAGE_DBG=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-records 2>&1 | grep 'stage\] sdp#A\|stage\] sdp#2\|ms_per_step' | tail -3 | sed 's/.*sdp#A *//; s/.*sdp#2 *//; s/.*"ms_per_step": \([0-9.]*\).*/step \1/' | tr '\n' ' ')"
done
cp /tmp/orig.so lra_amd/liblra_hip.so
