#!/bin/bash
# phase timelines of the update kernel (A/B build): variants given as crf_ab.sh specs; every spec writes gpurun_out/tl_<n>.bin
cd $GRAFT_REPO_ROOT
n=0
specs=()
for spec in "$@"; do
  n=$((n+1))
  specs+=("$spec WSC_CRF_UPD_TIMELINE=gpurun_out/tl_$n.bin AB_R=3")
done
bash profiles/crf_ab.sh "${specs[@]}"
for i in $(seq 1 $n); do python profiles/upd_timeline.py gpurun_out/tl_$i.bin; done
unset WSC_EXTRA_HIP_FLAGS
touch wsss-analysis_amd/csrc/crf.hip
python __graft_entry__.py > /dev/null 2>&1
