#!/usr/bin/env python
"""Per-kernel mean of a PMC counter from a rocprofv3 rocpd database (one --pmc pass).
    python tools/pmc_summary.py results.db [name-pattern]
"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
pat = sys.argv[2] if len(sys.argv) > 2 else "%"
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]


def tab(prefix):
    return [t for t in tabs if t.startswith(prefix)][0]


kd, ks, pe, pi = tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol"), tab("rocpd_pmc_event"), tab("rocpd_info_pmc")
if "--schema" in sys.argv:
    for t in (pe, pi):
        print(t, [r[1] for r in cur.execute("pragma table_info(%s)" % t)])
    sys.exit(0)
q = ("select s.kernel_name, d.grid_size_x, d.grid_size_y, p.name, count(*), avg(e.value), avg(d.end-d.start) "
     "from %s e join %s p on e.pmc_id = p.id join %s d on e.event_id = d.event_id "
     "join %s s on d.kernel_id = s.id where s.kernel_name like ? group by 1,2,3,4 order by 2*1,3,1"
     % (pe, pi, kd, ks))
for r in cur.execute(q, (pat,)):
    name = r[0].replace("_ZN2tk", "").split("EvP")[0][:40]
    print("%-42s grid=(%6d,%4d) %-12s n=%3d mean=%14.1f  dur=%8.1fus" % (name, r[1], r[2], r[3], r[4], r[5], r[6] / 1e3))
