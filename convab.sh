#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q -k "conv or net or edge" 2>&1 | tail -3
for a in resnet50 vgg16; do timeout 300 python bench.py --workload cam --arch $a --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['value'], {k:v for k,v in d['stages'].items() if k!='kernels'}, {k:(v['launches_per_step'],v.get('TFLOP/s')) for k,v in d['stages']['kernels'].items() if 'conv' in k})"; done
rm -rf gpurun_out/prof_conv
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof_conv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipeline --workload cam > gpurun_out/prof_conv.log 2>&1
python profiles/conv_layer_table.py gpurun_out/prof_conv/*/*_results.db > gpurun_out/layers_new.txt; tail -1 gpurun_out/layers_new.txt
