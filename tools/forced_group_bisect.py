#!/usr/bin/env python
"""Where do the milliseconds of the N > 1 path go on ONE rank?

Runs `bench.py` (configs[1], no kernel records) as a child per variant and prints ms/step:
the plain step, and the step with a forced one-rank process group under every combination of
(collective implementation) x (overlap slices) x (lab switches of parallel.FlatGradArena).
VERDICT r3 item 1: the forced group cost 110 -> 125 ms; this finds which part."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASE = dict(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1")

VARIANTS = [
    ("plain (no process group)", {}, []),
    ("forced PG, ProcessGroupNCCL async, 4 slices from hooks [r3 default]", dict(TK_FORCE_PROCESS_GROUP="1"), ["--overlap-buckets", "4"]),
    ("forced PG, ProcessGroupNCCL async, one all-reduce after backward", dict(TK_FORCE_PROCESS_GROUP="1"), ["--overlap-buckets", "0"]),
    ("forced PG, ProcessGroupNCCL on the current stream, one all-reduce", dict(TK_FORCE_PROCESS_GROUP="1", TK_ARENA_LAB="sync"), ["--overlap-buckets", "0"]),
    ("forced PG, arena issues nothing (group + watchdog exist)", dict(TK_FORCE_PROCESS_GROUP="1", TK_ARENA_LAB="noop"), ["--overlap-buckets", "0"]),
    ("forced PG, hooks installed, nothing issued", dict(TK_FORCE_PROCESS_GROUP="1", TK_ARENA_LAB="hooks"), ["--overlap-buckets", "4"]),
    ("forced PG, C-ABI RCCL side stream, 4 slices from hooks", dict(TK_FORCE_PROCESS_GROUP="1", TK_RCCL_DIRECT="1"), ["--overlap-buckets", "4"]),
    ("forced PG, C-ABI RCCL side stream, one all-reduce", dict(TK_FORCE_PROCESS_GROUP="1", TK_RCCL_DIRECT="1"), ["--overlap-buckets", "0"]),
    ("forced PG, C-ABI RCCL on the current stream, one all-reduce", dict(TK_FORCE_PROCESS_GROUP="1", TK_RCCL_DIRECT="1", TK_RCCL_INSTREAM="1"), ["--overlap-buckets", "0"]),
    ("forced PG, ProcessGroupNCCL async, 4 slices, 2 hardware queues", dict(TK_FORCE_PROCESS_GROUP="1", GPU_MAX_HW_QUEUES="2"), ["--overlap-buckets", "4"]),
    ("plain again (drift check)", {}, []),
]


def main():
    only = sys.argv[1:]
    port = 29700
    for k, (name, env, extra) in enumerate(VARIANTS):
        if only and str(k) not in only:
            continue
        e = dict(os.environ)
        if env:
            e.update(BASE)
            e["MASTER_PORT"] = str(port + k)
        e.update(env)
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline",
               "--no-pmc", "--no-rowk", "--no-kernel-records"] + extra
        pr = subprocess.run(cmd, env=e, capture_output=True, text=True)
        line = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
        if pr.returncode != 0 or not line:
            print("%2d %-75s FAILED rc=%d %s" % (k, name, pr.returncode, pr.stderr[-300:].replace("\n", " | ")), flush=True)
            continue
        d = json.loads(line[-1])
        r = d.get("rccl") or {}
        print("%2d %-75s %8.3f ms/step  %8.1f chunks/s  %s" % (
            k, name, d["ms_per_step"], d["value"],
            ("allreduce %.1f us, slices %s" % (r.get("allreduce_us", 0), r.get("bucket_us"))) if r else ""), flush=True)


if __name__ == "__main__":
    main()
