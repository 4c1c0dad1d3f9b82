"""Turn a rocprofv3 (ROCm 7.2) rocpd SQLite result into the per-kernel --stats table.

    python profiles/summarize_rocpd.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.txt
"""
import sqlite3
import sys


def main(path, top=40):
    c = sqlite3.connect(path)
    rows = c.execute(
        "SELECT S.display_name, COUNT(*), SUM(K.end-K.start), MIN(K.end-K.start), MAX(K.end-K.start), "
        "MAX(S.arch_vgpr_count), MAX(S.sgpr_count), MAX(S.group_segment_size) "
        "FROM rocpd_kernel_dispatch K JOIN rocpd_info_kernel_symbol S ON S.id = K.kernel_id AND S.guid = K.guid "
        "GROUP BY S.display_name ORDER BY 3 DESC").fetchall()
    total = sum(r[2] for r in rows)
    print("# %s" % path)
    print("# total kernel time %.3f ms over %d dispatches" % (total / 1e6, sum(r[1] for r in rows)))
    print("%-100s %8s %12s %10s %10s %10s %6s %5s %5s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us",
                                                            "pct", "vgpr", "sgpr", "lds"))
    for name, n, tot, mn, mx, vg, sg, lds in rows[:top]:
        print("%-100s %8d %12.1f %10.2f %10.2f %10.2f %6.2f %5s %5s %7s" % (name[:100], n, tot / 1e3, tot / n / 1e3,
                                                                          mn / 1e3, mx / 1e3, 100.0 * tot / total, vg,
                                                                          sg, lds))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
