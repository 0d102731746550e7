#!/usr/bin/env python
"""Shader-sequencer counters of kernel A's launches (instruction mix, wave cycles, issue stalls) from rocprofv3 PMC
passes -- each pass a separate run of `tools/crfbench.py` under `rocprofv3 --kernel-trace --pmc ...` (counters in
runs of their own, as the MI355X guide prescribes), read from the rocpd databases.

    python tools/sq_counters.py [--shapes cfg2r,rowK]
"""
import argparse
import collections
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = ["SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES", "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS",
          "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY", "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"]


def read(db):
    con = sqlite3.connect(db)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    tab = lambda p: [t for t in tabs if t.startswith(p)][0]
    kd, ks, pe, pi = tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol"), tab("rocpd_pmc_event"), tab("rocpd_info_pmc")
    q = ("select s.kernel_name, d.grid_size_x, p.name, e.value from %s e join %s d on e.event_id = d.event_id "
         "join %s s on d.kernel_id = s.id join %s p on e.pmc_id = p.id" % (pe, kd, ks, pi))
    return list(cur.execute(q))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="cfg2r,rowK")
    args = ap.parse_args()
    if shutil.which("rocprofv3") is None:
        raise SystemExit("rocprofv3 is not on PATH")
    acc = collections.defaultdict(lambda: [0.0, 0])
    for counters in PASSES:
        out = tempfile.mkdtemp(prefix="sq_", dir="/tmp")
        cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + counters.split() + ["-d", out, "-o", "sq", "--", sys.executable,
               os.path.join(ROOT, "tools", "crfbench.py"), "--shapes", args.shapes, "--modes", "band", "--reps", "3"]
        pr = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
        dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(out) for f in fs if f.endswith(".db")]
        if pr.returncode != 0 or not dbs:
            print("pass %s failed: %s" % (counters, (pr.stderr or pr.stdout)[-300:]))
            continue
        for name, grid, cname, val in read(dbs[0]):
            if "crf_band" in name:
                k = (name.replace("_ZN2tk", "").split("EvNS")[0][:44], int(grid), cname)
                acc[k][0] += float(val)
                acc[k][1] += 1
        shutil.rmtree(out, ignore_errors=True)
    last = None
    for (name, grid, cname), (v, n) in sorted(acc.items()):
        if (name, grid) != last:
            print("%s  grid.x %d" % (name, grid))
            last = (name, grid)
        print("    %-22s %14.4g per launch (%d launches)" % (cname, v / n, n))


if __name__ == "__main__":
    main()
