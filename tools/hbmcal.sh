# Counter calibration passes (run on the GPU box through gpurun): tools/hbmcal.sh [TAG]
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/hbmcal_f gpurun_out/hbmcal_w
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/hbmcal_f -o f --output-format csv -- tools/micro/hbmcal > gpurun_out/${tag}_hbmcal_stdout.txt 2> gpurun_out/hbmcal_f.err && echo calfetch ok && \
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/hbmcal_w -o w --output-format csv -- tools/micro/hbmcal > /dev/null 2> gpurun_out/hbmcal_w.err && echo calwrite ok && \
python3 tools/hbmcal_to_json.py gpurun_out/hbmcal_f gpurun_out/hbmcal_w gpurun_out/${tag}_hbmcal_stdout.txt $tag > gpurun_out/${tag}_hbmcal.json && \
cp profiles/hbm_counter_calibration.json gpurun_out/hbm_counter_calibration.json
find gpurun_out/hbmcal_f -name "*counter_collection.csv" -exec cp {} gpurun_out/${tag}_hbmcal_fetch.csv \;
find gpurun_out/hbmcal_w -name "*counter_collection.csv" -exec cp {} gpurun_out/${tag}_hbmcal_write.csv \;
