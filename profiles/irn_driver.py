"""BASELINE config 4 through its driver (make_sem_seg_labels.sem_seg_batches, the flavour bench.py times: ResNet50, f16x3,
16 images of 375 x 500 per batch): one JSON line; run under rocprofv3 --kernel-trace for profiles/busy_timeline.py."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "wsss-analysis_amd"))
import bench  # noqa: E402

print(json.dumps(bench.irn_measure(0, "f16x3", reps=int(os.environ.get("IRN_REPS", 8)))))
