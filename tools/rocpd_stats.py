#!/usr/bin/env python
"""Per-kernel summary (calls, total, mean, min, max in us) from a rocprofv3 rocpd
SQLite database -- the same numbers `rocprofv3 --stats` prints as kernel_stats.
    python tools/rocpd_stats.py results.db [top_n]
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    # optional: only the last FRAC of the trace's time span (steady state)
    frac = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    con = sqlite3.connect(db)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    lo, hi = list(cur.execute("select min(start), max(end) from %s" % kd))[0]
    cut = hi - (hi - lo) * frac
    q = ("select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), "
         "min(d.end-d.start), max(d.end-d.start) from %s d join %s s on d.kernel_id = s.id "
         "where d.start >= %d group by s.kernel_name order by 3 desc" % (kd, ks, cut))
    print("window: last %.0f%% of the trace = %.1f ms" % (100 * frac, (hi - cut) / 1e6))
    rows = list(cur.execute(q))
    tot = sum(r[2] for r in rows)
    print("%-90s %6s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "mean_us", "min_us",
                                                 "max_us", "%"))
    for name, n, s, a, mn, mx in rows[:top]:
        print("%-90s %6d %12.1f %10.1f %10.1f %10.1f %6.2f" % (name[:90], n, s / 1e3, a / 1e3,
                                                               mn / 1e3, mx / 1e3, 100.0 * s / tot))
    print("total kernel time: %.1f us over %d kernels" % (tot / 1e3, len(rows)))


if __name__ == "__main__":
    main()
