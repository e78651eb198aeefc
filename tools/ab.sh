#!/bin/bash
# A/B on one box: bench as built, then rebuild the named sources with -DWM_AB and bench again.
#   tools/ab.sh "synthesis" [bench args...]
set -e
names="$1"; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
# whatever happens (a bench run that times out, a JSON line that does not parse), the tree ends with the plain build
restore() { cd "$root/hts-train-world_amd/csrc" && for n in $names; do touch $n.hip; done && make -s > "$root/gpurun_out/ab_make2.log" 2>&1; }
trap restore EXIT
run() { timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-side --cpu-utts 2 "$@" > gpurun_out/ab_$tag.log 2>&1
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab_$tag.log").read().strip().splitlines()[-1])
print("$tag", d["ms_per_step"], {k: round(v, 3) for k, v in d["roofline"].get("kernel_ms_per_step", {}).items()})
PY
}
mkdir -p gpurun_out
tag=A; run "$@"
cd hts-train-world_amd/csrc
for n in $names; do touch $n.hip; done
make -s EXTRA=-DWM_AB > ../../gpurun_out/ab_make.log 2>&1
cd ../..
tag=B; run "$@"
tag=B2; run "$@"
restore
cd "$root"
tag=A2; run "$@"
