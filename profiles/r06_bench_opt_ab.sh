#!/bin/bash
# pipelined bench with a ctx option toggled: usage r06_bench_opt_ab.sh "<opt=val,...>" "<opt=val,...>" ...  (two rounds each)
cd $GRAFT_REPO_ROOT
for round in 1 2; do
for o in "$@"; do
  echo "#### WSC_BENCH_OPT=$o (round $round)"
  WSC_BENCH_OPT="$o" python bench.py --no-cpu-baseline --quick --steps 30 --warmup 4 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['stages']
print('value %.1f  ms_per_step %.3f  cnn %.3f  crf_create %.3f  crf_infer %.3f  sum %.3f' % (d['value'], d['ms_per_step'], s.get('cnn_ms', 0), s.get('crf_create_ms', 0), s.get('crf_infer_ms', 0), s.get('sum_ms', 0)))"
done
done
