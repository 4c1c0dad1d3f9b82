#!/bin/bash
# round 6: (1) what an LDS input window can gain at most (A/B build, WSC_CONV_DEBUG=32: the A pieces of a chunk's first tap only --
# results wrong, timing only) against what a 2-D output tile of 4 x 32 pixels costs in matrix work (the same layer on the map
# padded to multiples of 4 x 32), for the layers the linear window does not reach; (2) VGG16 per-dispatch SQ counter table.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out
cp wsss-analysis_amd/wsscam/libwsscam.so /tmp/lib_ship.so
trap 'cp /tmp/lib_ship.so wsss-analysis_amd/wsscam/libwsscam.so' EXIT
cp ab_tmp/libwsscam_ab.so wsss-analysis_amd/wsscam/libwsscam.so
{
echo "# layer: N Cin H W Cout k stride pad | base, A for the first tap only (window upper bound), the 4 x 32-quantised map"
for spec in "64 64 81 81 64 3 1 1|64 64 84 96 64 3 1 1" "32 64 321 321 64 3 1 1|32 64 324 352 64 3 1 1" "32 128 160 160 128 3 1 1|32 128 160 160 128 3 1 1" "32 256 80 80 256 3 1 1|32 256 80 96 256 3 1 1" "64 128 41 41 128 3 1 1|64 128 44 64 128 3 1 1" "64 256 21 21 256 3 1 1|64 256 24 32 256 3 1 1"; do
  IFS='|' read -r base quant <<< "$spec"
  WSC_BENCH_OPT=7=0 WSC_CONV_DEBUG=0 python profiles/conv_one.py $base f16x3 0 5 | tail -1
  WSC_BENCH_OPT=7=0 WSC_CONV_DEBUG=32 python profiles/conv_one.py $base f16x3 0 5 | tail -1
  WSC_BENCH_OPT=7=0 WSC_CONV_DEBUG=0 python profiles/conv_one.py $quant f16x3 0 5 | tail -1
done
} > $out/r06_conv_window_bound.txt 2>&1
cp /tmp/lib_ship.so wsss-analysis_amd/wsscam/libwsscam.so
[ "$1" = "bound" ] && { cat $out/r06_conv_window_bound.txt; exit 0; }
for i in 1 2; do
  [ $i = 1 ] && C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"
  [ $i = 2 ] && C="SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_ANY"
  rm -rf $out/pmcv_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex 'conv_igemm|cam_head_kernel' -d $out/pmcv_$i -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pipeline --quick --workload cam --arch vgg16 --batch 16 > $out/r06_pmcv_$i.log 2>&1
done
{ echo "# rocprofv3 --kernel-trace --pmc <SQ set 1 | SQ set 2> --kernel-include-regex 'conv_igemm|cam_head_kernel' -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pipeline --quick --workload cam --arch vgg16 --batch 16   (VGG16-CAM, 32 samples at 321^2, f16x3: one forward pass)"
  python profiles/conv_pmc_generic.py $out/pmcv_1/*/*_results.db $out/pmcv_2/*/*_results.db 40; } > $out/r06_pmc_conv_vgg16.txt 2>&1
rm -rf $out/pmcv_1 $out/pmcv_2
cat $out/r06_conv_window_bound.txt; tail -45 $out/r06_pmc_conv_vgg16.txt
