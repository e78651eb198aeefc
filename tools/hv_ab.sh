# Harvest pass, in-tree library against another build, alternating: tools/hv_ab.sh other.so
other="$(readlink -f "$1")"
for r in 1 2 3; do
  for lib in "" "$other"; do
    if [ -z "$lib" ]; then unset WORLD_MI355_LIB; tag=in-tree; else export WORLD_MI355_LIB=$lib; tag=$(basename $lib); fi
    python bench.py --workload harvest --steps 10 --warmup 3 --no-cpu-baseline --prewarm 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernel_ms_per_step']
print('$tag', d['ms_per_step'], {n: round(v,3) for n,v in k.items()})"
  done
done
