#!/bin/bash
# rebuild crf.hip with the fused Gaussian blur's two LDS variants on the box and time the no-pipeline bench
for v in 0 1 0 1; do
  export WSC_EXTRA_HIP_FLAGS="-DWSC_BLUR3_INPLACE=$v"
  touch wsss-analysis_amd/csrc/crf.hip
  python __graft_entry__.py > /dev/null 2>&1 || { echo "build failed $v"; continue; }
  echo "#### WSC_BLUR3_INPLACE=$v"
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipeline --quick 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages']
print('value',d['value'],'crf_infer_ms',s['crf_infer_ms'])
for k,v in s['kernels'].items():
    if 'update' in k or 'blur' in k: print('   ',k,v['launches_per_step'],v['avg_us'],v['ms_per_step'])
"
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --quick 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('pipelined value',d['value'])"
done
