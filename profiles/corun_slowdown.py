"""Which kernels of the pipelined step slow each other down?  From two kernel traces -- the pipelined step (conv stack i+1
beside mean-field loop i) and the --no-pipeline step (one stage at a time) -- the duration of every conv layer group and of
the update kernel alone and co-scheduled, and which layer group an update launch mostly ran beside.

    python profiles/corun_slowdown.py <pipelined_results.db> <no_pipeline_results.db>
"""
import sqlite3
import sys

import numpy as np


def load(path):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    return c.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)).fetchall()


def stacks(rows):
    out, cur = [], None
    for n, s, e in rows:
        if "stem_pool" in n:
            cur = [(s, e)]
        elif cur is not None and "conv_igemm" in n:
            cur.append((s, e))
        elif cur is not None and "cam_head" in n:
            cur.append((s, e))
            out.append(cur)
            cur = None
    return [st for st in out if len(st) == 53]


GROUPS = {"stem": range(0, 1), "layer1": range(1, 10), "layer2": range(10, 22), "layer3": range(22, 40), "layer4": range(40, 52),
          "head": range(52, 53)}
UPD = "update_splat_kernelILb1ELb1ELb1ELb1ELi4"
pipe, alone = load(sys.argv[1]), load(sys.argv[2])
sp, sa = stacks(pipe)[3:], stacks(alone)
da = np.array([[(e - s) / 1e3 for s, e in st] for st in sa]).mean(0)
dp = np.array([[(e - s) / 1e3 for s, e in st] for st in sp]).mean(0)
loop = [(s, e) for n, s, e in pipe if "update_splat" in n or "combine4" in n or "blur_lds" in n]


def beside(s, e, ivs):
    t = 0
    for a, b in ivs:
        if b <= s:
            continue
        if a >= e:
            break
        t += min(e, b) - max(s, a)
    return t / (e - s)


ov = np.array([[beside(s, e, loop) for s, e in st] for st in sp]).mean(0)
print("%d pipelined conv stacks, %d stand-alone" % (len(sp), len(sa)))
for g, r in GROUPS.items():
    r = list(r)
    print("%-7s alone %7.1f us   co-scheduled %7.1f us   x%.2f   fraction of its time beside loop kernels %.2f"
          % (g, da[r].sum(), dp[r].sum(), dp[r].sum() / da[r].sum(), np.average(ov[r], weights=dp[r])))
print("stack   alone %7.1f us   co-scheduled %7.1f us   x%.2f" % (da.sum(), dp.sum(), dp.sum() / da.sum()))
ua = np.mean([(e - s) / 1e3 for n, s, e in alone if UPD in n])
convs = sorted((s, e, i) for st in sp for i, (s, e) in enumerate(st))
byg = {}
for n, s, e in pipe:
    if UPD not in n:
        continue
    w = {}
    for a, b, i in convs:
        if b <= s:
            continue
        if a >= e:
            break
        g = [k for k, r in GROUPS.items() if i in r][0]
        w[g] = w.get(g, 0) + min(e, b) - max(s, a)
    g = "nothing" if not w else (max(w, key=w.get) if max(w.values()) >= 0.6 * (e - s) else "mixed")
    byg.setdefault(g, []).append((e - s) / 1e3)
print("update kernel alone %.1f us" % ua)
for g, v in sorted(byg.items()):
    print("update kernel mostly beside %-8s n %3d   %.1f us   x%.2f" % (g, len(v), np.mean(v), np.mean(v) / ua))
