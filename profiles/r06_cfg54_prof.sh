#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out
rm -rf $out/prof_irn $out/prof_hsn
timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof_irn -- python3 profiles/irn_driver.py > $out/r06_bench_irn.json 2> $out/r06_prof_irn.err
{ echo "# rocprofv3 --kernel-trace -- python3 profiles/irn_driver.py  (config 4 through make_sem_seg_labels.sem_seg_batches, two batches in flight)"; python profiles/busy_timeline.py $out/prof_irn/*/*_results.db; } > $out/r06_busy_timeline_irn.txt 2>&1
{ echo "# the same trace, per kernel"; python profiles/summarize_rocpd.py $out/prof_irn/*/*_results.db; } > $out/r06_kernel_stats_irn.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $out/prof_hsn -- python3 bench.py --workload hsn --arch vgg16 --batch 16 --steps 10 --warmup 2 --no-cpu-baseline > $out/r06_bench_hsn_prof.json 2> $out/r06_prof_hsn.err
{ echo "# rocprofv3 --kernel-trace -- python3 bench.py --workload hsn --arch vgg16 --batch 16 --steps 10 --warmup 2 --no-cpu-baseline  (config 5 through segment_adp, two batches in flight)"; python profiles/busy_timeline.py $out/prof_hsn/*/*_results.db; } > $out/r06_busy_timeline_hsn.txt 2>&1
{ echo "# the same trace, per kernel"; python profiles/summarize_rocpd.py $out/prof_hsn/*/*_results.db; } > $out/r06_kernel_stats_hsn.txt 2>&1
cat $out/r06_busy_timeline_irn.txt $out/r06_busy_timeline_hsn.txt; cat $out/r06_bench_irn.json; head -30 $out/r06_kernel_stats_irn.txt
