#!/bin/bash
# A/B of env settings on the no-pipeline bench: prints crf_infer_ms and kernel classes
for v in "$@"; do
  echo "== $v"
  env $v python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipeline --quick 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages']
print('value',d['value'],'crf_infer_ms',s['crf_infer_ms'],'create',s['crf_create_ms'])
for k,v in s['kernels'].items():
    if 'conv' not in k: print('   ',k,v['launches_per_step'],v['avg_us'],v['ms_per_step'])
"
done
