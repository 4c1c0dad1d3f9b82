#!/bin/bash
# configs 5 and 4 through their drivers by the runtime's hardware-queue count and lane count: bash profiles/r06_hwq.sh
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_hwq.txt
: > $out
for q in 4 8; do
for lanes in 2 3; do
  for rep in 1 2; do
  echo "== GPU_MAX_HW_QUEUES=$q lanes=$lanes rep $rep" >> $out
  GPU_MAX_HW_QUEUES=$q WSC_BENCH_HSN_LANES=$lanes timeout 400 python bench.py --workload hsn --arch vgg16 --batch 16 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('hsn value', d['value'], 'ms_per_step', d['ms_per_step'])" >> $out
  GPU_MAX_HW_QUEUES=$q WSC_BENCH_IRN_LANES=$lanes python - >> $out 2>&1 <<'PY'
import sys, os, json
sys.path.insert(0, 'wsss-analysis_amd'); sys.path.insert(0, '.')
import bench
print('irn', bench.irn_measure(0, 'f16x3', reps=6)["value"])
PY
  done
done
done
cat $out
