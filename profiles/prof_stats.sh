#!/bin/bash
# quick per-kernel stats of the bench: bash profiles/prof_stats.sh <tag> [bench args]
tag=$1; shift
out=gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $out/prof_$tag
timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof_$tag -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipeline --quick "$@" > $out/${tag}_prof.log 2>&1
python profiles/summarize_rocpd.py $out/prof_$tag/*/*_results.db 30 > $out/${tag}_stats.txt 2>&1
rm -rf $out/prof_$tag
cat $out/${tag}_stats.txt | cut -c1-60,100-200
