#!/bin/bash
# per-layer conv table of one build: bash profiles/conv_ab.sh <tag> [ENV=VAL ...]
tag=$1; shift
out=gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do export "$v"; done
rm -rf $out/prof_$tag
timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof_$tag -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipeline --quick --workload cam > $out/${tag}_prof.log 2>&1
python profiles/conv_layer_table.py $out/prof_$tag/*/*_results.db > $out/${tag}_conv.txt 2>&1
rm -rf $out/prof_$tag
cat $out/${tag}_conv.txt
