for v in 0 6 4 3 2; do
  export WORLD_MI355_SYN_SPLIT=$v
  timeout -k 10 300 python bench.py --workload synthesis --steps 5 --warmup 2 --prewarm 3 --no-cpu-baseline > gpurun_out/ss.json 2>gpurun_out/ss.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/ss.json").read().strip().splitlines()[-1])
print("split $v:", d["ms_per_step"], {k: round(x,2) for k,x in d["roofline"]["kernel_ms_per_step"].items()})
PY
done
