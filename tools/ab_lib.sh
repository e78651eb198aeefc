#!/bin/bash
# A/B of two builds on one box: the in-tree library (A) against another libworld_mi355.so (B), alternating.  The
# in-tree file is never touched: the other build is selected through WORLD_MI355_LIB (hts-train-world_amd/world.py).
#   tools/ab_lib.sh path/to/other.so [bench args...]
set -e
other="$(readlink -f "$1")"; shift
mkdir -p gpurun_out
run() { timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-side --cpu-utts 2 "$@" > gpurun_out/ab_$tag.log 2>&1
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab_$tag.log").read().strip().splitlines()[-1])
k=(d.get("roofline") or {}).get("kernel_ms_per_step", {})
print("$tag", d["ms_per_step"], {n: round(v, 3) for n, v in k.items() if v > 0.3})
PY
}
for round in 1 2 3; do
  unset WORLD_MI355_LIB; tag=A$round; run "$@"
  export WORLD_MI355_LIB="$other"; tag=B$round; run "$@"
done
