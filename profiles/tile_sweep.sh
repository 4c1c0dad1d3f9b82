#!/bin/bash
# rebuild crf.hip with different pixel-tile shapes on the box and time the no-pipeline bench
for wh in "16 16" "16 12" "12 12" "12 16" "20 16" "16 20" "8 32" ; do
  set -- $wh
  export WSC_EXTRA_HIP_FLAGS="-DWSC_TILE_W=$1 -DWSC_TILE_H=$2"
  touch wsss-analysis_amd/csrc/crf.hip
  python __graft_entry__.py > /dev/null 2>&1 || { echo "build failed $wh"; continue; }
  echo "#### TILE_W=$1 TILE_H=$2"
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipeline --quick 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages']
print('value',d['value'],'crf_infer_ms',s['crf_infer_ms'],'create',s['crf_create_ms'])
for k,v in s['kernels'].items():
    if 'update' in k or 'blur' in k or 'build' in k: print('   ',k,v['launches_per_step'],v['avg_us'],v['ms_per_step'])
"
done
