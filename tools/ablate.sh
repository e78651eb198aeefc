#!/bin/bash
# usage: tools/ablate.sh ENVVAR "v1 v2 ..." kernel_name [extra env]
var=$1; vals=$2; kern=$3
for m in $vals; do
  env $var=$m $4 timeout -k 10 400 python bench.py --steps 2 --warmup 1 --utts 128 --no-cpu-baseline 2>/dev/null > /tmp/ab.json
  python - "$m" "$kern" <<'PY'
import sys, json
d = json.load(open('/tmp/ab.json'))
print(sys.argv[1], sys.argv[2], d["roofline"]["kernel_ms_per_step"][sys.argv[2]], "total", d["ms_per_step"])
PY
done
