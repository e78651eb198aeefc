# Second SQ counter set (waits and memory instructions) per kernel on 64 utterances: tools/pmc_sq2.sh TAG [FS]
tag=${1:-r02}
fs=${2:-16000}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmcsq2_$tag
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d gpurun_out/pmcsq2_$tag -o sq --output-format csv -- python3 bench.py --fs $fs --steps 1 --warmup 0 --utts ${UTTS:-64} --no-cpu-baseline --prewarm 0 $BENCH_ARGS > gpurun_out/${tag}_pmc_sq2.json 2> gpurun_out/${tag}_pmc_sq2.err && echo sq2 ok
find gpurun_out/pmcsq2_$tag -name "*counter_collection.csv" -exec cp {} gpurun_out/${tag}_pmc_sq2.csv \;
python3 - <<PY
import csv, collections
rows = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open("gpurun_out/${tag}_pmc_sq2.csv")):
    n = r["Kernel_Name"].split("(")[0].replace("void wm::", "")[:44]
    rows[n][r["Counter_Name"]] += float(r["Counter_Value"])
print("%-46s %9s %10s %8s %8s %8s %8s %8s" % ("kernel", "waves", "cyc/wave", "wait%", "any%", "vmem%", "lds%", "vm r/w"))
for n, c in sorted(rows.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:14]:
    w = max(c["SQ_WAVES"], 1)
    cyc = max(c["SQ_WAVE_CYCLES"], 1)
    print("%-46s %9d %10.0f %7.1f%% %7.1f%% %7.1f%% %7.1f%% %5.0f/%-5.0f" % (n, w, cyc / w, 100 * c["SQ_WAIT_ANY"] / cyc, 100 * c["SQ_ACTIVE_INST_ANY"] / cyc,
          100 * c["SQ_ACTIVE_INST_VMEM"] / cyc, 100 * c["SQ_ACTIVE_INST_LDS"] / cyc, c["SQ_INSTS_VMEM_RD"] / w, c["SQ_INSTS_VMEM_WR"] / w))
PY
