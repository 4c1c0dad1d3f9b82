#!/bin/bash
# the three drivers with the lanes' conv stacks chained on the device (default) or interleaved by the hardware scheduler
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_chain_ab.txt
: > $out
for round in 1 2 3; do
for c in 1 0; do
  echo "== WSC_BENCH_CHAIN=$c (round $round)" >> $out
  WSC_BENCH_CHAIN=$c python bench.py --workload make_cam 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('make_cam', d['value'])" >> $out
  WSC_BENCH_CHAIN=$c timeout 400 python bench.py --workload hsn --arch vgg16 --batch 16 --steps 18 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('hsn', d['value'])" >> $out
  WSC_BENCH_CHAIN=$c python - >> $out 2>/dev/null <<'PY'
import sys, os, json
sys.path.insert(0, 'wsss-analysis_amd'); sys.path.insert(0, '.')
import torch
import bench
print('irn', bench.irn_measure(0, 'f16x3', reps=6)["value"])
PY
done
done
cat $out
