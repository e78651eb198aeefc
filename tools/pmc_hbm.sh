# FETCH_SIZE and WRITE_SIZE in separate passes (guide: TCC has 4 counters; FETCH_SIZE costs 3, WRITE_SIZE 2), 64 utterances
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o f --output-format csv -- python3 bench.py --steps 1 --warmup 0 --utts 64 --no-cpu-baseline > gpurun_out/pmc_fetch.log 2>&1 && echo fetch ok && \
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write -o w --output-format csv -- python3 bench.py --steps 1 --warmup 0 --utts 64 --no-cpu-baseline > gpurun_out/pmc_write.log 2>&1 && echo write ok
