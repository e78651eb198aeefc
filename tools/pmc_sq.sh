# SQ counters per kernel on 64 utterances (separate --pmc pass, kernel-trace only): instruction mix and issue/stall split.
#   tools/pmc_sq.sh TAG [FS]
tag=${1:-r02}
fs=${2:-16000}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmcsq_$tag
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d gpurun_out/pmcsq_$tag -o sq --output-format csv -- python3 bench.py --fs $fs --steps 1 --warmup 0 --utts ${UTTS:-64} --no-cpu-baseline --prewarm 0 $BENCH_ARGS > gpurun_out/${tag}_pmc_sq.json 2> gpurun_out/${tag}_pmc_sq.err && echo sq ok
find gpurun_out/pmcsq_$tag -name "*counter_collection.csv" -exec cp {} gpurun_out/${tag}_pmc_sq.csv \;
python3 - <<PY
import csv, collections
rows = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open("gpurun_out/${tag}_pmc_sq.csv")):
    n = r["Kernel_Name"].split("(")[0].replace("void wm::", "")[:44]
    rows[n][r["Counter_Name"]] += float(r["Counter_Value"])
print("%-46s %9s %9s %8s %8s %8s %7s %7s" % ("kernel", "waves", "valu/wave", "salu/w", "lds/w", "smem/w", "wait%", "valu%"))
for n, c in sorted(rows.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:14]:
    w = max(c["SQ_WAVES"], 1)
    cyc = max(c["SQ_WAVE_CYCLES"], 1)
    print("%-46s %9d %9.0f %8.0f %8.0f %8.0f %6.1f%% %6.1f%%" % (n, w, c["SQ_INSTS_VALU"] / w, c["SQ_INSTS_SALU"] / w, c["SQ_INSTS_LDS"] / w,
          c["SQ_INSTS_SMEM"] / w, 100 * c["SQ_WAIT_INST_ANY"] / cyc, 100 * c["SQ_ACTIVE_INST_VALU"] / cyc))
PY
