#!/bin/bash
# Collects every round artifact under profiles/ in ONE gpurun call:
#   gpurun --timeout 2400 -- 'bash profiles/collect.sh r04'
# Outputs land in gpurun_out/<tag>_* (merged back by gpurun); copy the ones to keep into profiles/.
tag=${1:-r06}
out=gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export WSC_PROFILE_ROUND=$tag
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $out/${tag}_pytest_gpu.txt
timeout 300 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
timeout 300 python bench.py --no-cpu-baseline --no-pipeline --quick > $out/${tag}_bench_nopipeline.json 2>> $out/${tag}_bench.err
timeout 300 python profiles/bench_irn.py > $out/${tag}_bench_irn_stages.json 2>> $out/${tag}_bench.err
# round 6: the same step with round 5's two-launch mean-field iteration (gauss_msg_kernel writes E, the update reads it)
WSC_BENCH_OPT=9=0 timeout 300 python bench.py --no-cpu-baseline --quick --steps 30 --warmup 4 > $out/${tag}_bench_two_launch.json 2>> $out/${tag}_bench.err
timeout 300 python bench.py --no-cpu-baseline --quick --steps 30 --warmup 4 > $out/${tag}_bench_msg_in_update.json 2>> $out/${tag}_bench.err
timeout 400 python bench.py --workload hsn --arch vgg16 --batch 16 --steps 10 --warmup 2 > $out/${tag}_bench_hsn.json 2>> $out/${tag}_bench.err
timeout 600 python bench.py --no-cpu-baseline --quick --seconds 30 > $out/${tag}_bench_30s.json 2>> $out/${tag}_bench.err
rm -rf $out/prof_stats $out/pmc_f $out/pmc_w
timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof_stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipeline --quick > $out/${tag}_prof_stats.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipeline --quick"; python profiles/summarize_rocpd.py $out/prof_stats/*/*_results.db; } > $out/${tag}_kernel_stats_bench.txt 2>&1
{ echo "# one ResNet50-CAM forward (64 samples @321^2, f16x3: the headline mode) out of the same trace"; python profiles/conv_layer_table.py $out/prof_stats/*/*_results.db 64 321 2; } > $out/${tag}_conv_layers.txt 2>&1
rm -rf $out/prof_f16
timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof_f16 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipeline --quick --precision f16 --workload cam > $out/${tag}_prof_f16.log 2>&1
{ echo "# the same in the fast f16 mode (bench.py --precision f16 --workload cam)"; python profiles/conv_layer_table.py $out/prof_f16/*/*_results.db 64 321 1; } > $out/${tag}_conv_layers_f16.txt 2>&1
RX='update_splat_kernel|gauss_msg_kernel|combine4_kernel|combine4_balanced_kernel|blur_lds_kernel|blur4_kernel|blur3_tile_kernel|conv_igemm|stem_pool_kernel|cam_head_kernel|tile_embed|slice_norm_tile|tile_slots|neighbors_kernel|slot_dest|scan_|slot_ones|assign_rows|first_bits|combine1_kernel|blur1_kernel|fill_tables'
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "$RX" -d $out/pmc_f -- python3 bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-pipeline --quick > $out/${tag}_pmc_f.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "$RX" -d $out/pmc_w -- python3 bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-pipeline --quick > $out/${tag}_pmc_w.log 2>&1
{ echo "# rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --kernel-include-regex '$RX' -- python3 bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-pipeline --quick"
  echo "# KiB per dispatch summed over the XCD instances; FETCH_SIZE must be doubled on gfx950 (MI355X_MICROARCH.md, HBM section)"
  python profiles/summarize_pmc.py $out/pmc_f/*/*_results.db $out/pmc_w/*/*_results.db; } > $out/${tag}_pmc_hbm_traffic.txt 2>&1
for i in 1 2; do
  [ $i = 1 ] && C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"
  [ $i = 2 ] && C="SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_ANY"
  rm -rf $out/pmcc_$i
  timeout 200 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex 'conv_igemm|stem_pool_kernel|cam_head_kernel' -d $out/pmcc_$i -- python3 bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-pipeline --quick --workload cam > $out/${tag}_pmcc_$i.log 2>&1
done
{ echo "# rocprofv3 --kernel-trace --pmc <SQ set 1 | SQ set 2> --kernel-include-regex conv_igemm -- python3 bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-pipeline --workload cam"; python profiles/conv_pmc_table.py $out/pmcc_1/*/*_results.db $out/pmcc_2/*/*_results.db; } > $out/${tag}_pmc_conv.txt 2>&1
for i in 1 2; do
  [ $i = 1 ] && C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU"
  [ $i = 2 ] && C="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM"
  rm -rf $out/pmcq_$i
  timeout 200 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex 'update_splat_kernel|gauss_msg_kernel|blur3_tile_kernel|combine4_kernel|combine4_balanced_kernel|blur_lds_kernel|blur4_kernel' -d $out/pmcq_$i -- python3 bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-pipeline --quick > $out/${tag}_pmcq_$i.log 2>&1
done
{ echo "# rocprofv3 --kernel-trace --pmc <SQ set 1 | SQ set 2> --kernel-include-regex 'update_splat_kernel|gauss_msg_kernel|blur3_tile_kernel|combine4_kernel|blur4_kernel' -- python3 bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-pipeline --quick"; python profiles/summarize_pmc.py $out/pmcq_1/*/*_results.db $out/pmcq_2/*/*_results.db; } > $out/${tag}_pmc_crf.txt 2>&1
rm -rf $out/prof_pipe
timeout 300 rocprofv3 --kernel-trace -d $out/prof_pipe -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --quick > $out/${tag}_prof_pipe.log 2>&1
{ echo "# rocprofv3 --kernel-trace -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --quick  (the PIPELINED step: three streams)"; python profiles/busy_timeline.py $out/prof_pipe/*/*_results.db; } > $out/${tag}_busy_timeline.txt 2>&1
python profiles/make_traffic_json.py $out/pmc_f/*/*_results.db $out/pmc_w/*/*_results.db > $out/${tag}_hbm_traffic.json 2>> $out/${tag}_bench.err
# round 6: configs 4 and 5 through their drivers under the kernel trace (busy timelines + per-kernel stats), VGG16 counter table
bash profiles/r06_cfg54_prof.sh > $out/${tag}_cfg54_prof.log 2>&1
bash profiles/r06_conv_probe.sh > $out/${tag}_conv_probe.log 2>&1
# round 6: the end-to-end leg alone under the kernel + memory-copy trace (busy fraction, per-stream timeline), and its rates
bash profiles/r06_e2e_prof.sh > $out/${tag}_e2e_prof.log 2>&1
E2E_REPS=3 timeout 200 python profiles/e2e_probe.py > $out/${tag}_e2e_probe.txt 2>&1
# round 6: the make_cam driver leg (config 1 through step.make_cam.run) under the kernel + memory-copy trace
bash profiles/r06_make_cam_prof.sh > $out/${tag}_make_cam_prof.log 2>&1
ls -la $out | grep ${tag}_
