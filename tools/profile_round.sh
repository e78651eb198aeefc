# Evidence set of a round (run on the GPU box through gpurun): tools/profile_round.sh TAG [FS]
#   1. rocprofv3 --kernel-trace --stats of the headline bench            -> gpurun_out/TAG_kernel_stats.csv (+ bench line)
#   2. separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes, 64 utterances -> gpurun_out/TAG_pmc_{fetch,write}.csv
#   3. profiles/pmc_traffic.json record for bench.py's roofline.traffic    -> gpurun_out/TAG_pmc_traffic.json
# Counters run with --kernel-trace only (no other trace domain), each pass in its own process.
#   tools/profile_round.sh TAG FS WORKLOAD   (WORKLOAD: analysis_synthesis (default), harvest, synthesis)
tag=${1:-r03}
fs=${2:-16000}
wl=${3:-analysis_synthesis}
if [ "$wl" = "analysis_synthesis" ]; then wa="--fs $fs --no-side"; u64="--utts 64"; else wa="--workload $wl"; u64=""; fi
if [ "$wl" = "synthesis" ]; then u64="--utts 256"; fi
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_$tag gpurun_out/pmcf_$tag gpurun_out/pmcw_$tag
# the stats pass times what the line times: no host-inclusive legs (their launches run beside copy kernels), the default
# pre-warm, the driver's step counts -- kernel_stats.csv's AverageNs of the dominant kernel is the line's launch_ms
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o ks --output-format csv -- python3 bench.py $wa --steps 20 --warmup 5 --no-cpu-baseline --no-host-inclusive --detail-file gpurun_out/${tag}_bench_under_rocprof_detail.json > gpurun_out/${tag}_bench_under_rocprof.json 2> gpurun_out/${tag}_rocprof.err && echo stats ok && \
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmcf_$tag -o f --output-format csv -- python3 bench.py $wa --steps 1 --warmup 0 $u64 --no-cpu-baseline --prewarm 0 > gpurun_out/${tag}_pmc_fetch.json 2> gpurun_out/${tag}_pmc_fetch.err && echo fetch ok && \
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmcw_$tag -o w --output-format csv -- python3 bench.py $wa --steps 1 --warmup 0 $u64 --no-cpu-baseline --prewarm 0 > gpurun_out/${tag}_pmc_write.json 2> gpurun_out/${tag}_pmc_write.err && echo write ok
find gpurun_out/prof_$tag -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_kernel_stats.csv \;
find gpurun_out/prof_$tag -name "*kernel_trace.csv" -exec cp {} gpurun_out/${tag}_kernel_trace.csv \;
find gpurun_out/pmcf_$tag -name "*counter_collection.csv" -exec cp {} gpurun_out/${tag}_pmc_fetch.csv \;
find gpurun_out/pmcw_$tag -name "*counter_collection.csv" -exec cp {} gpurun_out/${tag}_pmc_write.csv \;
frames=$(python3 -c "import json,sys; print(json.load(open('gpurun_out/${tag}_pmc_fetch.json'))['config']['frames_per_gpu'])")
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic_before.json 2>/dev/null
python3 tools/pmc_to_json.py gpurun_out/pmcf_$tag gpurun_out/pmcw_$tag $frames $fs $tag $wl && cp profiles/pmc_traffic.json gpurun_out/${tag}_pmc_traffic.json
