#!/bin/bash
# pieces / overlap sweep of synthesis_render on one box (configs[1] headline and configs[4])
mkdir -p gpurun_out
for ov in 1 0; do for pc in 1 2 4 8; do
  export WORLD_MI355_SYN_OVERLAP=$ov WORLD_MI355_SYN_PIECES=$pc
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-side --cpu-utts 2 > gpurun_out/sw_h.log 2>&1 || exit 1
  timeout -k 10 300 python bench.py --workload synthesis --steps 5 --warmup 2 --cpu-utts 2 > gpurun_out/sw_s.log 2>&1 || exit 1
  python - <<PY
import json
h=json.loads(open("gpurun_out/sw_h.log").read().strip().splitlines()[-1])
s=json.loads(open("gpurun_out/sw_s.log").read().strip().splitlines()[-1])
k=h["roofline"]["kernel_ms_per_step"]
print("overlap $ov pieces $pc: headline", h["ms_per_step"], "pulse", round(k["synth_pulse_kernel"],3), "ola", round(k["synth_ola_kernel"],3), "| synthesis", s["ms_per_step"], s["value"])
PY
done; done
