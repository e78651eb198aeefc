#!/bin/bash
# workgroups per resident slot of the persistent per-frame kernels (Context::oversub) on configs[1] and configs[4]
mkdir -p gpurun_out
for ov in ${OVS:-8 16 32 8 16 32 64}; do
  export WORLD_MI355_OVERSUB=$ov
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-side --cpu-utts 2 > gpurun_out/os_h.log 2>&1 || exit 1
  timeout -k 10 300 python bench.py --workload synthesis --steps 5 --warmup 2 --cpu-utts 2 > gpurun_out/os_s.log 2>&1 || exit 1
  python - <<PY
import json
h=json.loads(open("gpurun_out/os_h.log").read().strip().splitlines()[-1])
s=json.loads(open("gpurun_out/os_s.log").read().strip().splitlines()[-1])
k=h["roofline"]["kernel_ms_per_step"]
print("oversub $ov: headline", h["ms_per_step"], {n: round(v, 3) for n, v in k.items() if v > 2.0}, "| synthesis", s["ms_per_step"], s["value"])
PY
done
