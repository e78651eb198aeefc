#!/bin/bash
# A/B of one environment knob on one box, alternating: tools/ab_env.sh NAME VALUE_A VALUE_B [bench args...]
set -e
name="$1"; va="$2"; vb="$3"; shift 3
mkdir -p gpurun_out
run() { timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-side --no-cpu-baseline "$@" > gpurun_out/abe_$tag.log 2>&1
  python - <<PY
import json
d=json.loads(open("gpurun_out/abe_$tag.log").read().strip().splitlines()[-1])
k=(d.get("roofline") or {}).get("kernel_ms_per_step", {})
print("$tag", d["ms_per_step"], {n: round(v, 3) for n, v in k.items() if v > 0.3})
PY
}
for round in 1 2 3; do
  export $name="$va"; tag=A$round; run "$@"
  export $name="$vb"; tag=B$round; run "$@"
done
