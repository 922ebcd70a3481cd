#!/bin/bash
# Same-box A/B of two builds of the library: ab_base/liblra_hip.so (A) against lra_amd/liblra_hip.so (B), alternating, bench.py's own flags behind "--".
# usage (on the GPU box): bash tools/ab_bench.sh <tag> <pairs> -- <bench flags>
tag=$1; pairs=$2; shift 3
cp lra_amd/liblra_hip.so /tmp/ab_new.so
for i in $(seq 1 $pairs); do
  for v in A B; do
    if [ $v = A ]; then cp ab_base/liblra_hip.so lra_amd/liblra_hip.so; else cp /tmp/ab_new.so lra_amd/liblra_hip.so; fi
    python bench.py "$@" > gpurun_out/${tag}_${v}${i}.json 2> gpurun_out/${tag}_${v}${i}.err
    python - <<P
import json
for l in open("gpurun_out/${tag}_${v}${i}.json"):
    if l.startswith("{"):
        d = json.loads(l); k = d["kernel_ms_per_step"]
        print("$v$i", round(d["value"], 4), round(d["ms_per_step"], 1), d["cpu_baseline"].get("sample_equals_gpu") if "cpu_baseline" in d else None,
              {x: round(k[x], 1) for x in ("sdp_process", "sdp_process_wg", "sdp_build", "ir_fill") if x in k})
P
  done
done
cp /tmp/ab_new.so lra_amd/liblra_hip.so
