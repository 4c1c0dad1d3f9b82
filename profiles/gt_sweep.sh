#!/bin/bash
# rebuild crf.hip with different interior tiles of the fused Gaussian blur (halo 2 on each side) and time the bench
for wh in "12 12" "28 12" "12 28" "20 12" "12 20" "12 12"; do
  set -- $wh
  export WSC_EXTRA_HIP_FLAGS="-DWSC_GTI=$1 -DWSC_GTJ=$2"
  touch wsss-analysis_amd/csrc/crf.hip
  python __graft_entry__.py > /dev/null 2>&1 || { echo "build failed $wh"; continue; }
  echo "#### GTI=$1 GTJ=$2"
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipeline --quick 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages']
print('value',d['value'],'crf_infer_ms',s['crf_infer_ms'])
for k,v in s['kernels'].items():
    if 'blur' in k: print('   ',k,v['launches_per_step'],v['avg_us'],v['ms_per_step'])
"
done
unset WSC_EXTRA_HIP_FLAGS; touch wsss-analysis_amd/csrc/crf.hip; python __graft_entry__.py > /dev/null 2>&1
