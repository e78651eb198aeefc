# one guarded attempt at the write-side HBM counter on a tiny workload (the pass hung on this pool at 64 utterances)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_wr
timeout -k 10 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_wr -o wr --output-format csv -- python3 bench.py --steps 1 --warmup 0 --utts 8 --no-cpu-baseline > gpurun_out/pmc_wr.log 2>&1
echo rc=$?
ls gpurun_out/pmc_wr 2>/dev/null | head
