#!/bin/bash
# configs 5 and 4 through their drivers (two batches in flight), plus their tests
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_net.py tests/test_gpu_irn.py tests/test_gpu_hsn.py -m gpu -q -x 2>&1 | tail -4
timeout 400 python bench.py --workload hsn --arch vgg16 --batch 16 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('hsn value', d['value'], 'ms_per_step', d['ms_per_step'])"
python - <<'PY'
import sys, os, json
sys.path.insert(0, 'wsss-analysis_amd'); sys.path.insert(0, '.')
import bench
print('irn', json.dumps(bench.irn_measure(0, 'f16x3', reps=6)))
PY
