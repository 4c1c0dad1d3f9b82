#!/bin/bash
# SQ counters of the lattice-build kernels: bash profiles/pmc_build.sh <tag>
tag=$1; shift
out=gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
RX='tile_embed_kernel|tile_slots_kernel|slot_ones_kernel|neighbors_kernel|pack_pixels_kernel|flag_first_kernel|remap_kernel|slice_norm_kernel|tile_scale_entries|assign_ids|slot_dest|scan_'
for i in 1 2; do
  [ $i = 1 ] && C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
  [ $i = 2 ] && C="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM"
  rm -rf $out/pmcb_$i
  timeout 200 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex "$RX" -d $out/pmcb_$i -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pipeline --quick > $out/${tag}_pmcb_$i.log 2>&1
done
python profiles/summarize_pmc.py $out/pmcb_1/*/*_results.db $out/pmcb_2/*/*_results.db > $out/${tag}_pmc_build.txt 2>&1
rm -rf $out/pmcb_1 $out/pmcb_2
cat $out/${tag}_pmc_build.txt
