"""GPU busy fraction of the pipelined bench out of a rocprofv3 --kernel-trace database: over the last `n` steps (a step ends
with its last update_splat launch... simpler: the whole window between the first and the last dispatch of the second half of
the trace), the union of all kernel intervals against the wall span, and the largest idle gaps with the kernels around them.

    rocprofv3 --kernel-trace -d gpurun_out/prof_pipe -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --quick
    python profiles/busy_timeline.py gpurun_out/prof_pipe/*/*_results.db
"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = c.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)).fetchall()
# the timed region = the densest part: take the window from 40 % to 80 % of the dispatches (steady pipelined steps)
lo, hi = int(len(rows) * 0.40), int(len(rows) * 0.80)
win = rows[lo:hi]
t0, t1 = win[0][1], max(r[2] for r in win)
busy, cur_s, cur_e = 0, None, None
gaps = []
prev_name = ""
for name, s, e in win:
    if cur_e is None:
        cur_s, cur_e = s, e
    elif s <= cur_e:
        cur_e = max(cur_e, e)
    else:
        gaps.append((s - cur_e, prev_name, name, cur_e - t0))
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    prev_name = name
busy += cur_e - cur_s
short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:44]
n_upd = sum(1 for r in win if "update_splat_kernel" in r[0])
print("window %.2f ms, %d dispatches (~%.1f steps), GPU busy %.2f ms = %.1f %%, idle %.2f ms in %d gaps"
      % ((t1 - t0) / 1e6, len(win), n_upd / 11.0, busy / 1e6, 100.0 * busy / (t1 - t0), (t1 - t0 - busy) / 1e6, len(gaps)))
over = sum((e - s) for _, s, e in win) - busy
print("sum of kernel durations %.2f ms (overlap between streams %.2f ms)" % (sum((e - s) for _, s, e in win) / 1e6, over / 1e6))
for g, a, b, at in sorted(gaps, reverse=True)[:12]:
    print("  gap %7.1f us at %8.2f ms  after %-44s before %s" % (g / 1e3, at / 1e6, short(a), short(b)))
hist = [0, 0, 0, 0]
for g, _, _, _ in gaps:
    hist[0 if g < 2e3 else 1 if g < 10e3 else 2 if g < 50e3 else 3] += g
print("idle by gap size: <2us %.2f ms, 2-10us %.2f ms, 10-50us %.2f ms, >50us %.2f ms" % tuple(h / 1e6 for h in hist))
