"""Per-kernel averages of rocprofv3 --pmc counters from rocpd SQLite results (values summed over
the per-XCD/SE instances of a dispatch, then averaged over dispatches of the same kernel).

    python profiles/summarize_pmc.py gpurun_out/pmc1/p_*_results.db
"""
import sqlite3
import sys
from collections import defaultdict


def main(paths):
    agg = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for path in paths:
        c = sqlite3.connect(path)
        per = defaultdict(float)
        meta = {}
        for name, disp, d, cn, cv in c.execute("select name, dispatch_id, duration, counter_name, counter_value from pmc_events"):
            per[(name, disp, cn)] += cv
            meta[(name, disp)] = d
        for (name, disp, cn), v in per.items():
            agg[name][cn].append(v)
        for (name, disp), d in meta.items():
            dur[name].append(d)
    for name in sorted(agg, key=lambda n: -sum(dur[n])):
        short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        print("%s   dispatches=%d  avg_us(profiled)=%.1f" % (short, len(dur[name]), sum(dur[name]) / len(dur[name]) / 1e3))
        for cn in sorted(agg[name]):
            v = agg[name][cn]
            print("    %-32s avg %.4g   max %.4g" % (cn, sum(v) / len(v), max(v)))


if __name__ == "__main__":
    main(sys.argv[1:])
