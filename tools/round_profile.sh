# end-of-round evidence: kernel-trace stats of the headline bench + the four bench lines (tag = $1)
tag=${1:-r01_f}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_$tag
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o ks --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_bench_under_rocprof.json 2>gpurun_out/${tag}_rocprof.err && echo rocprof ok && \
timeout -k 10 500 python bench.py --steps 5 --warmup 2 > gpurun_out/${tag}_bench_config1.json 2>/dev/null && echo config1 ok && \
timeout -k 10 500 python bench.py --workload harvest --steps 3 --warmup 1 > gpurun_out/${tag}_bench_harvest_config3.json 2>/dev/null && echo harvest ok && \
timeout -k 10 300 python bench.py --workload synthesis --steps 5 --warmup 2 > gpurun_out/${tag}_bench_synthesis_config5.json 2>/dev/null && echo synthesis ok && \
timeout -k 10 300 python bench.py --workload codec --steps 3 --warmup 1 > gpurun_out/${tag}_bench_codec.json 2>/dev/null && echo codec ok
ls gpurun_out/prof_$tag | head
