cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_lds
timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d gpurun_out/pmc_lds -o lds --output-format csv -- python3 bench.py --steps 1 --warmup 0 --utts 64 --no-cpu-baseline --prewarm 0 > gpurun_out/pmc_lds.log 2>&1
echo rc=$?
ls -R gpurun_out/pmc_lds | head
