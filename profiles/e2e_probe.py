"""End-to-end leg of bench.py alone (host batch in, .npy files out): float32 input vs decoded uint8 input, several repetitions;
E2E_ONLY=u8|f32 runs one of them (for rocprofv3 --kernel-trace + busy_timeline.py)."""
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "wsss-analysis_amd"))
import bench  # noqa: E402

wl = bench.Workload(0, 32, "f16x3", "cam_crf", seed=0)
tmp = tempfile.mkdtemp(prefix="wsc_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
only = os.environ.get("E2E_ONLY", "")
K = int(os.environ.get("E2E_STEPS", 40))
try:
    for rep in range(int(os.environ.get("E2E_REPS", 2))):
        for u8 in (False, True):
            if only and (only == "u8") != u8:
                continue
            wl.setup_e2e(tmp, n_writers=int(os.environ.get("E2E_WRITERS", 8)), u8=u8)
            for _ in range(int(os.environ.get("E2E_WARM", 3))):
                wl.step_e2e()
            wl.drain_e2e()
            t0 = time.perf_counter()
            for _ in range(K):
                wl.step_e2e()
            wl.drain_e2e()
            wl.ctx.sync()
            dt = time.perf_counter() - t0
            print("u8=%d  %.1f images/s  %.3f ms/step" % (u8, 32 * K / dt, dt / K * 1e3), flush=True)
            tr = wl.e2e.pop("trace", None)
            if tr:  # WSC_BENCH_E2E_TRACE=1: host time of a step by blocking call (includes the warm-up steps)
                n = tr.pop("steps")
                print("   host ms/step: " + "  ".join("%s %.3f" % (k, v / n * 1e3) for k, v in tr.items()), flush=True)
    # reference point: the resident-input pipelined step
    for _ in range(0 if os.environ.get("E2E_NO_RESIDENT") else 4):
        wl.step_pipelined()
    wl.drain()
    t0 = time.perf_counter()
    for _ in range(0 if os.environ.get("E2E_NO_RESIDENT") else K):
        wl.step_pipelined()
    wl.drain()
    wl.ctx.sync()
    dt = time.perf_counter() - t0
    if not os.environ.get("E2E_NO_RESIDENT"):
        print("resident input: %.1f images/s  %.3f ms/step" % (32 * K / dt, dt / K * 1e3))
finally:
    for k in ("pool", "finisher", "camcopier"):
        wl.e2e[k].shutdown(wait=True)
    shutil.rmtree(tmp, ignore_errors=True)
