#!/bin/bash
# The round's evidence set on one box: tools/evidence_r04.sh TAG   (every step writes under gpurun_out/)
tag=${1:-r04_d}
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh ${tag} 16000 > gpurun_out/${tag}_profile.log 2>&1; tail -2 gpurun_out/${tag}_profile.log
bash tools/profile_round.sh ${tag}48 48000 > gpurun_out/${tag}48_profile.log 2>&1; tail -2 gpurun_out/${tag}48_profile.log
bash tools/profile_round.sh ${tag}_hv 48000 harvest > gpurun_out/${tag}_hv_profile.log 2>&1; tail -2 gpurun_out/${tag}_hv_profile.log
bash tools/profile_round.sh ${tag}_syn 16000 synthesis > gpurun_out/${tag}_syn_profile.log 2>&1; tail -2 gpurun_out/${tag}_syn_profile.log
cp profiles/pmc_traffic.json gpurun_out/${tag}_pmc_traffic_all.json
