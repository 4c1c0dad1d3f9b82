"""Why is stages.value_hsn of the default bench line (first hsn_measure of a process: 810-930 images/s) below a later call
in the same process (1060-1130)?  Per-batch times of the timed segment_adp call, first call against third call."""
import argparse
import os
import sys
import time

import torch  # FIRST, as bench.py does: its bundled HIP runtime is then the process's only one (the library binds to it by soname)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "wsss-analysis_amd"))
import bench  # noqa: E402
from wsscam.hsn import demo as hsn_demo  # noqa: E402

ha = argparse.Namespace(batch=16, steps=int(os.environ.get("PROBE_STEPS", 9)), warmup=1, no_cpu_baseline=True, precision="f16x3")
orig = hsn_demo.run_batches_on_lanes
log = []


def timed_run(n_batches, ctxs, one_batch):
    t0 = time.perf_counter()
    rec = []

    def wrapped(b, ctx):
        a = time.perf_counter()
        r = one_batch(b, ctx)
        rec.append((b, ctxs.index(ctx), (a - t0) * 1e3, (time.perf_counter() - t0) * 1e3))
        return r

    out = orig(n_batches, ctxs, wrapped)
    log.append(sorted(rec))
    return out


hsn_demo.run_batches_on_lanes = timed_run
for call in range(3):
    log.clear()
    v = bench.hsn_measure(ha, 0)["value"]
    print("call %d: %.1f images/s" % (call, v))
    timed = [r for r in log if len(r) == ha.steps][0]
    print("   batch(lane) start..end ms: " + "  ".join("%d(%d) %.0f..%.0f" % r for r in timed), flush=True)
