#!/bin/bash
# SQ counter table of the conv stack: bash profiles/conv_pmc.sh <tag>   (the same passes as collect.sh runs)
tag=$1; shift
out=gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do
  [ $i = 1 ] && C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"
  [ $i = 2 ] && C="SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_ANY"
  rm -rf $out/pmcc_$i
  timeout 200 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex 'conv_igemm|stem_pool_kernel|cam_head_kernel' -d $out/pmcc_$i -- python3 bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-pipeline --quick --workload cam > $out/${tag}_pmcc_$i.log 2>&1
done
{ echo "# rocprofv3 --kernel-trace --pmc <SQ set 1 | SQ set 2> --kernel-include-regex 'conv_igemm|stem_pool_kernel|cam_head_kernel' -- python3 bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-pipeline --quick --workload cam"; python profiles/conv_pmc_table.py $out/pmcc_1/*/*_results.db $out/pmcc_2/*/*_results.db; } > $out/${tag}_pmc_conv.txt 2>&1
rm -rf $out/pmcc_1 $out/pmcc_2
cat $out/${tag}_pmc_conv.txt
