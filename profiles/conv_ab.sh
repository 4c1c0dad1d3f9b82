#!/bin/bash
# Ablations of one conv layer with the A/B build: gpurun -- 'bash profiles/conv_ab.sh "<layer args>" [precision] [debug values]'
# WSC_CONV_DEBUG bits: 1 no DMA after the prologue, 2 no fragment reads / MFMAs (non-rolling loops), 4 no MFMAs,
# 8 no barrier in the rolling loop, 16 no fragment reads in the rolling loop (results are wrong by design)
cp wsss-analysis_amd/wsscam/libwsscam.so /tmp/lib_ship.so
trap 'cp /tmp/lib_ship.so wsss-analysis_amd/wsscam/libwsscam.so' EXIT  # a timeout / kill must not leave the A/B library in the package
cp ab_tmp/libwsscam_ab.so wsss-analysis_amd/wsscam/libwsscam.so
L=${1:-"64 512 21 21 512 3 1 1"}
P=${2:-f16x3}
D=${3:-"0 1 4 5"}
for d in $D; do WSC_CONV_DEBUG=$d python profiles/conv_one.py $L $P 0 5 | tail -1; done
for t in 1 512; do WSC_CONV_TILE=$t python profiles/conv_one.py $L $P 0 5 | tail -1; done
