# Effective shader clock per kernel (MI355X_MICROARCH.md, DVFS give-back: GRBM_GUI_ACTIVE / 8 XCDs / kernel time):
#   tools/pmc_clock.sh TAG [FS]      BENCH_ARGS selects another workload.  One --pmc pass with --kernel-trace only.
tag=${1:-r03}
fs=${2:-16000}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmcclk_$tag
timeout -k 10 400 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d gpurun_out/pmcclk_$tag -o clk --output-format csv -- python3 bench.py --fs $fs --steps 3 --warmup 1 --no-cpu-baseline --no-side --prewarm 5 $BENCH_ARGS > gpurun_out/${tag}_pmc_clock.json 2> gpurun_out/${tag}_pmc_clock.err && echo clock ok
find gpurun_out/pmcclk_$tag -name "*counter_collection.csv" -exec cp {} gpurun_out/${tag}_pmc_clock.csv \;
python3 - <<PY
import csv, collections
rows = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open("gpurun_out/${tag}_pmc_clock.csv")):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
        continue
    n = r["Kernel_Name"].split("(")[0].replace("void wm::", "")[:48]
    dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    if dur < 3e5:                      # shorter than 0.3 ms: the quotient reads high (guide)
        continue
    rows[n][0] += float(r["Counter_Value"]); rows[n][1] += dur; rows[n][2] += 1
print("%-50s %6s %10s %9s" % ("kernel", "calls", "avg ms", "GHz"))
for n, (cyc, ns, k) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:16]:
    print("%-50s %6d %10.3f %9.3f" % (n, k, ns / k / 1e6, cyc / 8.0 / ns))
PY
