#!/bin/bash
# A/B of two builds on one box: the in-tree library (A) against another libworld_mi355.so (B), alternating.
#   tools/ab_lib.sh path/to/other.so [bench args...]
set -e
other="$1"; shift
lib=hts-train-world_amd/libworld_mi355.so
cp $lib /tmp/lib_A.so
run() { timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-side --cpu-utts 2 "$@" > gpurun_out/ab_$tag.log 2>&1
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab_$tag.log").read().strip().splitlines()[-1])
k=d["roofline"].get("kernel_ms_per_step", {})
print("$tag", d["ms_per_step"], {n: round(v, 3) for n, v in k.items() if v > 1.0})
PY
}
for round in 1 2 3; do
  cp /tmp/lib_A.so $lib; tag=A$round; run "$@"
  cp "$other" $lib; tag=B$round; run "$@"
done
cp /tmp/lib_A.so $lib
