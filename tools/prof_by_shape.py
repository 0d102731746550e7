#!/usr/bin/env python
"""Kernel mean/min duration grouped by (kernel, grid) from a rocprofv3 rocpd database."""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
pat = sys.argv[2] if len(sys.argv) > 2 else "%"
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
q = ("select s.kernel_name, d.grid_size_x, d.grid_size_y, d.workgroup_size_x, count(*), "
     "avg(d.end-d.start), min(d.end-d.start) from %s d join %s s on d.kernel_id=s.id "
     "where s.kernel_name like ? group by 1,2,3 order by 2*1,3,1" % (kd, ks))
for r in cur.execute(q, (pat,)):
    name = r[0].replace("_ZN2tk", "").split("EvP")[0][:44]
    print("%-46s grid=(%6d,%4d) wg=%4d calls=%4d mean=%9.1fus min=%9.1fus"
          % (name, r[1], r[2], r[3], r[4], r[5] / 1e3, r[6] / 1e3))
