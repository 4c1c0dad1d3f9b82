#!/bin/bash
L=wsss-analysis_amd/wsscam/libwsscam.so
cp $L ab/orig.so
for r in 1 2 3; do
for n in A B C; do
  cp ab/lib_$n.so $L
  echo "== $n $(timeout 200 python bench.py --workload cam --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['stages']['cnn_ms'], d['stages']['conv_stack_tflops'])")"
done
done
cp ab/orig.so $L
