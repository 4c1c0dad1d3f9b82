"""Timeline of ONE bilateral lattice build (wsc_crf_create, 32 images at 321 x 321) out of a rocprofv3 --kernel-trace database:
kernel name, start offset, duration, gap to the previous kernel's end (host round trips show up as gaps).

    rocprofv3 --kernel-trace -d gpurun_out/prof_build -- python3 profiles/gbuild_time.py
    python profiles/gbuild_timeline.py gpurun_out/prof_build/*/*_results.db
"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = c.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)).fetchall()
# the last build: from the last fill_tables / fill_u64 kernel on
starts = [i for i, r in enumerate(rows) if "fill_tables_kernel" in r[0] or "fill_u64_kernel" in r[0]]
i0 = starts[-1]
t0 = rows[i0][1]
prev_end = t0
tot_k = 0.0
for name, s, e in rows[i0:]:
    short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:48]
    print("%-50s start %8.1f us  dur %7.1f us  gap %7.1f us" % (short, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    prev_end = e
    tot_k += (e - s) / 1e3
print("kernels %.1f us, span %.1f us" % (tot_k, (prev_end - t0) / 1e3))
