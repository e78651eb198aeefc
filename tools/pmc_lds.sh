cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_lds
timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d gpurun_out/pmc_lds -o lds --output-format csv -- python3 bench.py --steps 1 --warmup 0 --utts 64 --no-cpu-baseline --prewarm 0 ${2:+--fs $2} ${3:+--workload $3} > gpurun_out/pmc_lds.log 2>&1
echo rc=$?
ls -R gpurun_out/pmc_lds | head
find gpurun_out/pmc_lds -name "*counter_collection.csv" -exec cp {} gpurun_out/${1:-r04}_pmc_lds.csv \;
python3 - <<PY
import csv, collections
rows = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open("gpurun_out/${1:-r04}_pmc_lds.csv")):
    n = r["Kernel_Name"].split("(")[0].replace("void wm::", "")[:44]
    rows[n][r["Counter_Name"]] += float(r["Counter_Value"])
print("%-46s %10s %8s %10s %10s %10s" % ("kernel", "cyc/wave", "lds/w", "ldsact%", "ldswait%", "conflict%"))
for n, c in sorted(rows.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:12]:
    cyc = max(c["SQ_WAVE_CYCLES"], 1)
    print("%-46s %10s %8s %9.1f%% %9.1f%% %9.1f%%" % (n, "-", "-", 100 * c["SQ_ACTIVE_INST_LDS"] / cyc, 100 * c["SQ_WAIT_INST_LDS"] / cyc,
          100 * c["SQ_LDS_BANK_CONFLICT"] / max(1.0, c["SQ_LDS_IDX_ACTIVE"])))
PY
