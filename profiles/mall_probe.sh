#!/bin/bash
# Does a smaller batch keep the layer1/layer2 activations in the 256 MB Infinity Cache?  Per-layer conv times at 8/16/32 images.
out=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for b in 4 8 16 32; do
  rm -rf $out/mall_$b
  timeout 300 rocprofv3 --kernel-trace -d $out/mall_$b -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipeline --quick --workload cam --batch $b > $out/mall_$b.log 2>&1
  python profiles/conv_layer_table.py $out/mall_$b/*/*_results.db $((2*b)) 321 > $out/mall_layers_$b.txt 2>&1
  rm -rf $out/mall_$b
done
