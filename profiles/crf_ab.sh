#!/bin/bash
# usage: crf_ab.sh "<label>|<extra hipcc flags for crf.hip>|<env assignments>" ...   (rebuilds crf.hip when the flags change)
last="__none__"
for spec in "$@"; do
  IFS='|' read -r label flags envs <<< "$spec"
  if [ "$flags" != "$last" ]; then
    export WSC_EXTRA_HIP_FLAGS="$flags"
    touch wsss-analysis_amd/csrc/crf.hip
    python __graft_entry__.py > /dev/null 2>&1 || { echo "build failed: $flags"; continue; }
    last="$flags"
  fi
  echo "#### $label   [flags: $flags] [env: $envs]"
  env $envs python profiles/crf_ab.py 2>&1 | grep -v amdgpu.ids
done
