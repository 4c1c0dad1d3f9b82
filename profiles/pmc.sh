#!/bin/bash
# bash profiles/pmc.sh <tag> "<counters>" "<kernel regex>" [bench args]
tag=$1; C=$2; RX=$3; shift 3
out=gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $out/pmc_$tag
timeout 300 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex "$RX" -d $out/pmc_$tag -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pipeline --quick "$@" > $out/${tag}_pmc.log 2>&1
python profiles/summarize_pmc.py $out/pmc_$tag/*/*_results.db > $out/${tag}_pmc.txt 2>&1
rm -rf $out/pmc_$tag
cat $out/${tag}_pmc.txt | cut -c1-70,100-300
