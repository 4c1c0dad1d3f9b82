#!/bin/bash
# tiled random-walk step with / without the XCD-contiguous block order (A/B build), stage times of profiles/bench_irn.py
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_irn.py -q -m gpu -x 2>&1 | tail -2
cp ab_tmp/libwsscam_ab.so wsss-analysis_amd/wsscam/libwsscam.so
for r in 1 2; do for x in 1 0; do
  echo "WSC_RW_XCD=$x: $(WSC_RW_XCD=$x python profiles/bench_irn.py --arch resnet50 --batch 8 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('random walk %.3f ms/image (batch of 32), single image %.3f ms, driver %.1f images/s' % (d['random_walk_ms_per_image'], d['random_walk_ms_single_image_call'], d['driver_images_per_s']))")"
done; done
