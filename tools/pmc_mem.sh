# Memory-side counters per kernel on 64 utterances (L1->L2 read request latency, L2 hit rate; more counters than these four do not fit one pass): tools/pmc_mem.sh TAG [FS]
tag=${1:-r02}
fs=${2:-16000}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmcmem_$tag
timeout -k 10 150 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCC_HIT TCC_MISS -d gpurun_out/pmcmem_$tag -o m --output-format csv -- python3 bench.py --fs $fs --steps 1 --warmup 0 --utts ${UTTS:-64} --no-cpu-baseline --prewarm 0 $BENCH_ARGS > gpurun_out/${tag}_pmc_mem.json 2> gpurun_out/${tag}_pmc_mem.err && echo mem ok
find gpurun_out/pmcmem_$tag -name "*counter_collection.csv" -exec cp {} gpurun_out/${tag}_pmc_mem.csv \;
python3 - <<PY
import csv, collections
rows = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open("gpurun_out/${tag}_pmc_mem.csv")):
    n = r["Kernel_Name"].split("(")[0].replace("void wm::", "")[:44]
    rows[n][r["Counter_Name"]] += float(r["Counter_Value"])
print("%-46s %12s %9s %12s %9s %8s %10s" % ("kernel", "rd req", "rd lat", "wr req", "wr lat", "L2 hit%", "tlb miss%"))
for n, c in sorted(rows.items(), key=lambda kv: -kv[1].get("TCP_TCC_READ_REQ", 0))[:12]:
    rr = max(c["TCP_TCC_READ_REQ"], 1); wr = max(c["TCP_TCC_WRITE_REQ"], 1)
    print("%-46s %12.3e %9.0f %12.3e %9.0f %7.1f%% %9.2f%%" % (n, rr, c["TCP_TCC_READ_REQ_LATENCY"] / rr, wr, c["TCP_TCC_WRITE_REQ_LATENCY"] / wr,
          100 * c["TCC_HIT"] / max(c["TCC_HIT"] + c["TCC_MISS"], 1), 100 * c["TCP_UTCL1_TRANSLATION_MISS"] / max(c["TCP_UTCL1_REQUEST"], 1)))
PY
