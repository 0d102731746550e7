#!/usr/bin/env python
"""Start / end of the last launches in a rocprofv3 --kernel-trace database (rocpd), relative to the first of them:
which kernels ran beside which, and the gaps between them.

    python tools/kernel_timeline.py <db> [--last 24] [--like %tk%]"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--last", type=int, default=24)
    ap.add_argument("--like", default="%")
    args = ap.parse_args()
    cur = sqlite3.connect(args.db).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    tab = lambda p: [t for t in tabs if t.startswith(p)][0]     # noqa: E731
    kd, ks = tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol")
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % kd)]
    qcol = "d.queue_id" if "queue_id" in cols else "0"
    rows = list(cur.execute("select s.kernel_name, d.start, d.end, %s from %s d join %s s on d.kernel_id = s.id "
                            "where s.kernel_name like ? order by d.start" % (qcol, kd, ks), (args.like,)))[-args.last:]
    t0 = rows[0][1]
    prev_end = {}
    for name, st, en, q in rows:
        gap = (st - prev_end[q]) / 1e3 if q in prev_end else float("nan")
        print("q%-3s %9.1f -> %9.1f us  (%6.1f us, gap %6.1f)  %s" % (q, (st - t0) / 1e3, (en - t0) / 1e3, (en - st) / 1e3, gap, name[:60]))
        prev_end[q] = en


if __name__ == "__main__":
    main()
