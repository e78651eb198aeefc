#!/bin/bash
# Socket power and clocks while the headline step runs in a loop: tools/power_probe.sh [bench args...]
# (is the FP64-dense path held below the peak clock by the power cap?)
mkdir -p gpurun_out
python bench.py --steps 3000 --warmup 5 --prewarm 10 --no-side --no-cpu-baseline --no-host-inclusive "$@" > gpurun_out/power_probe_bench.json 2> gpurun_out/power_probe_bench.err &
pid=$!
sleep 24
for i in $(seq 1 12); do
  rocm-smi --showpower --showclocks --showuse --showtemp --json 2>/dev/null | tr -d '\n' ; echo
  sleep 1
done > gpurun_out/power_probe_smi.txt
wait $pid
rocm-smi --showmaxpower 2>&1 | head -30 > gpurun_out/power_probe_caps.txt
python - <<'PY'
import json
rows=[]
for l in open("gpurun_out/power_probe_smi.txt"):
    l=l.strip()
    if not l.startswith("{"): continue
    try: d=json.loads(l)
    except ValueError: continue
    for card,v in d.items():
        if not isinstance(v,dict): continue
        rows.append({k:v[k] for k in v if any(s in k.lower() for s in ("power","sclk","mclk","use","temperature (sensor junction)","fclk"))})
for r in rows[:12]: print(r)
PY
cat gpurun_out/power_probe_caps.txt
tail -c 400 gpurun_out/power_probe_bench.json
