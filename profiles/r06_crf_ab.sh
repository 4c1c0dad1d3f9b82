#!/bin/bash
# round 6, CRF A/B in one gpurun call: two-launch form vs message-in-update, combine variants
cd $GRAFT_REPO_ROOT
K="-DWSC_AB_KNOBS"
bash profiles/crf_ab.sh \
  "r5 form: gauss_msg + update, balanced combine|$K|AB_OPTS=9:0 WSC_CRF_COMBINE_BALANCED=1" \
  "msg in update, balanced combine|$K|AB_OPTS=9:1 WSC_CRF_COMBINE_BALANCED=1" \
  "msg in update, chunk 8 / 32 rows|$K|AB_OPTS=9:1" \
  "B=16 two launches|$K|AB_OPTS=9:0 AB_B=16 WSC_CRF_COMBINE_BALANCED=1" \
  "B=16 msg in update|$K|AB_OPTS=9:1 AB_B=16 WSC_CRF_COMBINE_BALANCED=1" \
  "M=5 two launches|$K|AB_OPTS=9:0 AB_M=5 WSC_CRF_COMBINE_BALANCED=1" \
  "M=5 msg in update|$K|AB_OPTS=9:1 AB_M=5 WSC_CRF_COMBINE_BALANCED=1" \
  "chunk 4 / 32 rows|$K -DWSC_CB_CHUNK=4|AB_OPTS=9:1"
# leave the shipped build behind
unset WSC_EXTRA_HIP_FLAGS
touch wsss-analysis_amd/csrc/crf.hip
python __graft_entry__.py > /dev/null 2>&1
