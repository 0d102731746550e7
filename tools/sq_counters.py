#!/usr/bin/env python
"""Shader-sequencer counters of kernel A's launches (instruction mix, wave cycles, issue stalls) from rocprofv3 PMC
passes -- each pass a separate run of `tools/crfbench.py` under `rocprofv3 --kernel-trace --pmc ...` (counters in
runs of their own, as the MI355X guide prescribes), read from the rocpd databases.  One shape per run, so that a
kernel's launches belong to a known shape; values are summed over the counter's instances (32 per launch) and
averaged over the launches.

    python tools/sq_counters.py [--shapes cfg2r,rowK] [--save profiles/r4_sq_counters.json]

The JSON (with the kernel-source hash) is what bench.py's `roofline_crf.*.issue_floor_us` is computed from."""
import argparse
import collections
import hashlib
import json
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = ["SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES", "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS",
          "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY", "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"]
# crfbench shape -> the (op, T, N, realistic chunk length) key bench.py uses
KEYS = {"cfg2r": "crf:800:128:4000", "rowK": "crf:4000:256:0", "cfg5r": "crf:1600:64:8000", "cfg4r": "catmod:800:128:4000"}


def kernel_hash():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "taiyaki_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(d, name), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def read(db):
    con = sqlite3.connect(db)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    tab = lambda p: [t for t in tabs if t.startswith(p)][0]     # noqa: E731
    kd, ks, pe, pi = tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol"), tab("rocpd_pmc_event"), tab("rocpd_info_pmc")
    q = ("select s.kernel_name, d.id, p.name, sum(e.value) from %s e join %s d on e.event_id = d.event_id "
         "join %s s on d.kernel_id = s.id join %s p on e.pmc_id = p.id group by 1, 2, 3" % (pe, kd, ks, pi))
    return list(cur.execute(q))


def role(name):
    if "crf_band_sweep" in name:
        return "sweep"
    if "crf_band_posterior" in name:
        return "posterior"
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="cfg2r,rowK")
    ap.add_argument("--save", default=None)
    ap.add_argument("--passes", default=None, help="other counter passes, ';' between passes (e.g. 'SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS')")
    args = ap.parse_args()
    global PASSES
    if args.passes:
        PASSES = [x.strip() for x in args.passes.split(";") if x.strip()]
    if shutil.which("rocprofv3") is None:
        raise SystemExit("rocprofv3 is not on PATH")
    doc = dict(kernel_hash=kernel_hash(), tool="tools/sq_counters.py", passes=PASSES, shapes={},
               note="per launch, summed over the counters' 32 instances; *_CYCLES of waves and ACTIVE / WAIT are "
                    "quad-cycles, SQ_BUSY_CYCLES cycles per instance (max over instances here)")
    for sh in args.shapes.split(","):
        acc = collections.defaultdict(list)
        names = {}
        for counters in PASSES:
            out = tempfile.mkdtemp(prefix="sq_", dir="/tmp")
            cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + counters.split() + ["-d", out, "-o", "sq", "--", sys.executable,
                   os.path.join(ROOT, "tools", "crfbench.py"), "--shapes", sh, "--modes", "bandnf", "--reps", "3"]
            pr = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
            dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(out) for f in fs if f.endswith(".db")]
            if pr.returncode != 0 or not dbs:
                print("pass %s failed: %s" % (counters, (pr.stderr or pr.stdout)[-300:]))
                continue
            for name, _disp, cname, val in read(dbs[0]):
                r = role(name)
                if r:
                    acc[(r, cname)].append(float(val))
                    names[r] = name
            shutil.rmtree(out, ignore_errors=True)
        rec = {}
        for (r, cname), vals in sorted(acc.items()):
            rec.setdefault(r, dict(kernel=names[r]))[cname] = sum(vals) / len(vals)
        doc["shapes"][KEYS.get(sh, sh)] = rec
        for r, d in rec.items():
            print("%s  %s" % (sh, d["kernel"][:70]))
            for k, v in d.items():
                if k != "kernel":
                    print("    %-22s %14.5g per launch" % (k, v))
            if "SQ_ACTIVE_INST_ANY" in d and "SQ_WAVE_CYCLES" in d:
                print("    waves issuing for %.1f %% of their resident time; VALU %.1f %% of the chip's VALU issue slots "
                      "over the kernel's busy cycles" % (100 * d["SQ_ACTIVE_INST_ANY"] / d["SQ_WAVE_CYCLES"],
                                                         100 * d.get("SQ_INSTS_VALU", 0) * 4 / (1024 * max(1.0, d.get("SQ_BUSY_CYCLES", 0) / 32))))
    if args.save:
        with open(args.save, "w") as fh:
            json.dump(doc, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
