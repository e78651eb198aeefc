for spec in "4 0.65" "5 0.7" "6 0.7" "4 0.65" "5 0.7" "6 0.75"; do
  set -- $spec
  export WORLD_MI355_SWEEP_TAPER=$2
  timeout -k 10 300 python bench.py --workload sweep --steps 3 --warmup 1 --no-cpu-baseline --rounds $1 > gpurun_out/sk.json 2>gpurun_out/sk.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/sk.json").read().strip().splitlines()[-1])
print("rounds $1 taper $2:", d["ms_per_step"], d["value"], {k: v for k, v in d["phases_ms_per_step"].items() if k != "note"})
PY
done
