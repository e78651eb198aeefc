#!/bin/bash
# configs[3] on one GPU over the writer-thread and round counts: tools/sweep_knobs.sh
cd $GRAFT_REPO_ROOT 2>/dev/null || true
python - <<PY
import os, importlib
sh = importlib.import_module("hts-train-world_amd.sharding")
print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "usable", sh.usable_cpus())
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    try: print(p, open(p).read().strip())
    except OSError as e: print(p, "-")
PY
for spec in "16 4" "32 4" "64 4" "8 4" "32 8" "32 16" "16 8"; do
  set -- $spec
  timeout -k 10 300 python bench.py --workload sweep --steps 3 --warmup 1 --no-cpu-baseline --io-threads $1 --rounds $2 > gpurun_out/sk.json 2>gpurun_out/sk.err || { tail -3 gpurun_out/sk.err; continue; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/sk.json").read().strip().splitlines()[-1])
print("threads $1 rounds $2:", d["ms_per_step"], "ms", d["phases_ms_per_step"], d["host_side"]["files_per_s"])
PY
done
