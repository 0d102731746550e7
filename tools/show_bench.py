#!/usr/bin/env python
"""The numbers of a bench.py JSON line that the docs quote.   python tools/show_bench.py <file>"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.1f %s  %.2f ms/step  roofline frac %.4f (%.1f us)" % (d["value"], d["unit"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["mean_us"]))
if "loss_path" in d:
    print("loss_path", {k: v for k, v in d["loss_path"].items() if k.endswith("_ms")})
if "roofline_crf" in d:
    for k in ("in_step", "rowK"):
        r = d["roofline_crf"].get(k)
        if r:
            print("crf %s: %.1f us (min %.1f) issue floor %.1f current=%s gated %s" % (k, r["mean_us"], r["min_us"], r.get("issue_floor_us", -1), r.get("issue_floor", {}).get("current"), r.get("gated_reads")))
if "roofline_in_step" in d:
    print("logz in step %.2f us" % d["roofline_in_step"]["mean_us"])
if d.get("rccl"):
    print("rccl overhead_ms", d["rccl"].get("overhead_ms"))
