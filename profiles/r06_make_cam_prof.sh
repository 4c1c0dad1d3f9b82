#!/bin/bash
# the make_cam driver leg (BASELINE config 1 through step.make_cam.run) under the kernel + memory-copy trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out
rm -rf $out/prof_mc
WSC_BENCH_MAKE_CAM_ONE_RUN=1 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $out/prof_mc -- python3 bench.py --workload make_cam > $out/r06_bench_make_cam.json 2> $out/r06_prof_mc.err
{ echo "# WSC_BENCH_MAKE_CAM_ONE_RUN=1 rocprofv3 --kernel-trace --memory-copy-trace -- python3 bench.py --workload make_cam   (a 64-image warm-up run, then ONE run of 1824 images)"; python profiles/busy_timeline.py $out/prof_mc/*/*_results.db; } > $out/r06_busy_timeline_make_cam.txt 2>&1
python profiles/step_timeline.py $out/prof_mc/*/*_results.db 0.7 30 > $out/r06_step_timeline_make_cam.txt 2>&1
cat $out/r06_bench_make_cam.json; head -14 $out/r06_busy_timeline_make_cam.txt; head -70 $out/r06_step_timeline_make_cam.txt
