#!/bin/bash
# kernel trace of the end-to-end leg alone and of the resident-input pipelined step: busy fraction + per-kernel durations
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out
rm -rf $out/prof_e2e $out/prof_res
E2E_ONLY=${E2E_ONLY:-f32} E2E_REPS=1 E2E_NO_RESIDENT=1 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $out/prof_e2e -- python3 profiles/e2e_probe.py > $out/r06_e2e_prof.log 2>&1
{ echo "# rocprofv3 --kernel-trace --memory-copy-trace -- python3 profiles/e2e_probe.py  (E2E_ONLY=f32 E2E_REPS=1 E2E_NO_RESIDENT=1)"; python profiles/busy_timeline.py $out/prof_e2e/*/*_results.db; } > $out/r06_busy_timeline_e2e.txt 2>&1
{ echo "# the same trace, per kernel"; python profiles/summarize_rocpd.py $out/prof_e2e/*/*_results.db; } > $out/r06_kernel_stats_e2e.txt 2>&1
python - <<'PY' >> $out/r06_busy_timeline_e2e.txt 2>&1
import glob, sqlite3
c = sqlite3.connect(glob.glob("gpurun_out/prof_e2e/*/*_results.db")[0])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
mc = [t for t in tabs if "memory_copy" in t]
print("# memory copies:", mc)
for t in mc:
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % t)]
    print(cols)
    rows = c.execute("select * from %s" % t).fetchall()
    print(len(rows), "rows")
    si, ei, zi = cols.index("start"), cols.index("end"), cols.index("size")
    by = {}
    for r in rows:
        k = r[zi]
        by.setdefault(k, []).append(r[ei] - r[si])
    for k, v in sorted(by.items(), key=lambda kv: -kv[0])[:12]:
        print("size %12d  n %5d  avg %9.1f us  %6.1f GB/s" % (k, len(v), sum(v) / len(v) / 1e3, k / (sum(v) / len(v))))
PY
python profiles/step_timeline.py $out/prof_e2e/*/*_results.db 0.6 40 > $out/r06_step_timeline_e2e.txt 2>&1; tail -3 $out/r06_e2e_prof.log; head -16 $out/r06_busy_timeline_e2e.txt; cat $out/r06_step_timeline_e2e.txt
