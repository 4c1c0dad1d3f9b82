#!/bin/bash
# blur_lds_kernel with fewer classes per workgroup (more, smaller workgroups): bench stage times per forced variant
cd $GRAFT_REPO_ROOT
export WSC_EXTRA_HIP_FLAGS="-DWSC_AB_KNOBS"
touch wsss-analysis_amd/csrc/crf.hip
python __graft_entry__.py > /dev/null 2>&1 || echo build failed
for round in 1 2; do
for v in 0 3 4; do
  echo "#### WSC_BLUR_LDS_MINVAR=$v (round $round)"
  WSC_BLUR_LDS_MINVAR=$v python bench.py --no-cpu-baseline --quick --no-pipeline --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['stages']; k = s['kernels']
print('ms_per_step %.3f crf_infer %.3f' % (d['ms_per_step'], s['crf_infer_ms']), {n: v['avg_us'] for n, v in k.items() if 'blur' in n or 'update' in n})"
done
done
unset WSC_EXTRA_HIP_FLAGS
touch wsss-analysis_amd/csrc/crf.hip
python __graft_entry__.py > /dev/null 2>&1
