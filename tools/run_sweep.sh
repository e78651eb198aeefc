#!/bin/bash
# Run length of Synthesis (WORLD_MI355_SYN_RUN, samples per run) on one box: headline and Synthesis-only passes.
set -e
mkdir -p gpurun_out
for S in 128 256 512 1024 2048; do
  for W in analysis_synthesis synthesis; do
    WORLD_MI355_SYN_RUN=$S timeout -k 10 300 python bench.py --workload $W --steps 10 --warmup 2 --no-side --no-cpu-baseline > gpurun_out/runs_$S_$W.log 2>&1
    python - <<PY
import json
d=json.loads(open("gpurun_out/runs_$S_$W.log").read().strip().splitlines()[-1])
k=d["roofline"]["kernel_ms_per_step"]
print("S=$S $W", d["ms_per_step"], {n: round(v,3) for n,v in k.items() if n in ("synth_pulse_kernel","synth_ola_kernel","d4c_kernel","cheaptrick_kernel")})
PY
  done
done
