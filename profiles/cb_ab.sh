#!/bin/bash
# usage: cb_ab.sh "<label>|<extra hipcc flags for crf.hip>" ...  -> class times of bench --no-pipeline
last="__none__"
for spec in "$@"; do
  IFS='|' read -r label flags <<< "$spec"
  if [ "$flags" != "$last" ]; then
    export WSC_EXTRA_HIP_FLAGS="$flags"
    touch wsss-analysis_amd/csrc/crf.hip
    python __graft_entry__.py > /dev/null 2>&1 || { echo "build failed: $flags"; continue; }
    last="$flags"
  fi
  echo "#### $label [flags: $flags]"
  python bench.py --no-cpu-baseline --quick --no-pipeline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['stages']['kernels']; print(d['ms_per_step'], d['stages']['crf_infer_ms'], [v for n, v in k.items() if n.startswith('combine4')][0])"
done
