#!/bin/bash
timeout 600 python -m pytest tests -m gpu -x -q -k "crf or edge" 2>&1 | tail -2
for i in 1 2 3; do
  timeout 200 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['stages']['kernels']
print(d['value'], d['stages']['crf_create_ms'], d['stages']['crf_infer_ms'])"
done
