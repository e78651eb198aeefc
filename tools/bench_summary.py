import json,sys
for f in sys.argv[1:]:
    d=json.loads(open(f).read().strip().splitlines()[-1]); s=d["summary"]
    print(f, {k:s[k] for k in ("ms_per_step","host_inclusive","host_inclusive_coded","harvest_ms","synthesis_ms","sweep_ms")})
    print("   ", d["side_workloads"]["sweep"]["phases_ms_per_step"])
