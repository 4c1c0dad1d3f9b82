"""Per-dispatch SQ counter table of the conv kernels of ANY stack from two rocprofv3 --pmc passes (same sets as conv_pmc_table.py):
    python profiles/conv_pmc_generic.py <set1_results.db> <set2_results.db> [n_last] [label,label,...]
Rows = the last n_last conv_igemm / stem_pool / cam_head dispatches of the trace in launch order (one network pass); labels
(optional, comma separated) name the rows.  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs)."""
import sqlite3
import sys


def load(db):
    c = sqlite3.connect(db)
    d = {}
    for disp, name, dur, cn, cv in c.execute("select dispatch_id, name, duration, counter_name, counter_value from pmc_events"):
        e = d.setdefault(disp, {"name": name, "dur": dur})
        e[cn] = e.get(cn, 0) + cv
    return [d[k] for k in sorted(d)]


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    if "conv_igemm_kernel<" in n:
        t = n.split("conv_igemm_kernel<")[1].split(">")[0].replace(" ", "").split(",")
        return "igemm %sx%s fast%s wpt%s" % (t[0], t[1], t[8] if len(t) > 8 else "0", t[9] if len(t) > 9 else "0")
    return n.split("(")[0][:28]


def main():
    a, b = load(sys.argv[1]), load(sys.argv[2])
    n_last = int(sys.argv[3]) if len(sys.argv) > 3 else len(a)
    labels = sys.argv[4].split(",") if len(sys.argv) > 4 else []
    a, b = a[-n_last:], b[-n_last:]
    print("%-3s %-34s %-26s %8s %9s %6s %6s" % ("#", "layer", "kernel", "us", "mfma_util", "wait%", "bank%"))
    busy = tot = 0.0
    for i, (x, y) in enumerate(zip(a, b)):
        us = x["dur"] / 1e3
        util = x.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (x["dur"] * 1e-9 * 2.4e9 * 1024)
        wait = 100.0 * y.get("SQ_WAIT_ANY", 0) / max(x.get("SQ_WAVE_CYCLES", 1), 1)
        bank = 100.0 * y.get("SQ_LDS_BANK_CONFLICT", 0) / max(y.get("SQ_LDS_IDX_ACTIVE", 1), 1)
        busy += x.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
        tot += x["dur"]
        print("%-3d %-34s %-26s %8.1f %8.1f%% %6.1f %6.1f" % (i, labels[i] if i < len(labels) else "", short(x["name"]), us, 100 * util, wait, bank))
    print("# %d dispatches, %.1f us, matrix pipe busy %.1f %% of SIMD-cycles" % (len(a), tot / 1e3, 100 * busy / (tot * 1e-9 * 2.4e9 * 1024)))


if __name__ == "__main__":
    main()
