"""Condensed per-stream timeline of a rocprofv3 --kernel-trace --memory-copy-trace database: one line per run of same-named
kernels on a stream (first start, last end, count), every memory copy, from the first stem_pool launch after `frac` of the trace
for `span_ms`.

    python profiles/step_timeline.py <results.db> [frac=0.6] [span_ms=40]
"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
span = float(sys.argv[3]) if len(sys.argv) > 3 else 40.0
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
mc = [t for t in tabs if "memory_copy" in t]
rows = c.execute("select s.kernel_name, d.start, d.end, d.stream_id from %s d join %s s on d.kernel_id = s.id" % (kd, ks)).fetchall()
ev = [(s, e, n.replace("_ZN12_GLOBAL__N_1", "")[:26], "K", st) for n, s, e, st in rows]
if mc:
    for r in c.execute("select start, end, size, stream_id, src_agent_id, dst_agent_id from %s" % mc[0]):
        ev.append((r[0], r[1], "COPY %.1f MB %s" % (r[2] / 1e6, "in" if r[4] < r[5] else "out"), "C", r[3]))
ev.sort()
ker = [e for e in ev if e[3] == "K"]
t0 = ker[int(len(ker) * frac)][0]
idx = next(i for i, e in enumerate(ev) if e[0] > t0 and "stem_pool" in e[2])
base = ev[idx][0]
runs = {}  # stream -> [name, start, end, count]


def flush(st):
    r = runs.pop(st, None)
    if r:
        out.append((r[1], "%9.1f %9.1f  st%-2s %-28s x%d" % ((r[1] - base) / 1e3, (r[2] - base) / 1e3, st, r[0], r[3])))


out = []
for s, e, name, kind, st in ev[idx:]:
    if s - base > span * 1e6:
        break
    cls = name[:12] if kind == "K" else name
    if "conv_igemm" in name:
        cls = "conv_igemm"
    r = runs.get(st)
    if r and r[0] == cls and kind == "K" and s - r[2] < 100e3:
        r[2], r[3] = max(r[2], e), r[3] + 1
    else:
        flush(st)
        runs[st] = [cls, s, e, 1]
for st in list(runs):
    flush(st)
for _, line in sorted(out):
    print(line)
