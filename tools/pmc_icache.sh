# Instruction-cache counters per kernel on 64 utterances: tools/pmc_icache.sh TAG [FS]
tag=${1:-r02}
fs=${2:-16000}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmcic_$tag
timeout -k 10 150 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_IFETCH -d gpurun_out/pmcic_$tag -o ic --output-format csv -- python3 bench.py --fs $fs --steps 1 --warmup 0 --utts ${UTTS:-64} --no-cpu-baseline --prewarm 0 $BENCH_ARGS > gpurun_out/${tag}_pmc_ic.json 2> gpurun_out/${tag}_pmc_ic.err && echo ic ok
find gpurun_out/pmcic_$tag -name "*counter_collection.csv" -exec cp {} gpurun_out/${tag}_pmc_ic.csv \;
python3 - <<PY
import csv, collections
rows = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open("gpurun_out/${tag}_pmc_ic.csv")):
    n = r["Kernel_Name"].split("(")[0].replace("void wm::", "")[:44]
    rows[n][r["Counter_Name"]] += float(r["Counter_Value"])
print("%-46s %12s %12s %8s %12s %14s" % ("kernel", "ic req", "ic miss", "miss%", "ifetch", "wave cyc"))
for n, c in sorted(rows.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:14]:
    print("%-46s %12.3e %12.3e %7.1f%% %12.3e %14.3e" % (n, c["SQC_ICACHE_REQ"], c["SQC_ICACHE_MISSES"], 100 * c["SQC_ICACHE_MISSES"] / max(c["SQC_ICACHE_REQ"], 1),
          c["SQ_IFETCH"], c["SQ_WAVE_CYCLES"]))
PY
