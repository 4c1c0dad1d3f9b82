#!/bin/bash
# pipelined step by conv tile policy (A/B build): does a smaller-LDS conv tile co-run better with the mean-field loop?
cd $GRAFT_REPO_ROOT
cp ab_tmp/libwsscam_ab.so wsss-analysis_amd/wsscam/libwsscam.so
out=gpurun_out/r06_tile_corun.txt
: > $out
for round in 1 2; do
for t in 0 1 -1 256; do
  echo "#### WSC_CONV_TILE=$t (round $round)" >> $out
  WSC_CONV_TILE=$t python bench.py --no-cpu-baseline --quick --steps 30 --warmup 4 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['stages']
print('value %.1f  ms_per_step %.3f  cnn %.3f  crf_create %.3f  crf_infer %.3f  sum %.3f' % (d['value'], d['ms_per_step'], s.get('cnn_ms', 0), s.get('crf_create_ms', 0), s.get('crf_infer_ms', 0), s.get('sum_ms', 0)))" >> $out
done
done
cat $out
