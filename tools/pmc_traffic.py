#!/usr/bin/env python
"""HBM traffic of the loss-path operators from rocprofv3 PMC passes (MI355X_MICROARCH.md, HBM section).

    python tools/pmc_traffic.py --ops logz:4000:256:0,crf:800:128:4000 [--json] [--save profiles/r2]

For every op spec `name:T:N:realistic_chunk_len` (name = logz | crf | catmod) the tool runs the
operator a few times under `rocprofv3 --pmc FETCH_SIZE` and again under `--pmc WRITE_SIZE` (the two
counters do not fit one pass; --pmc is never combined with any trace domain but --kernel-trace),
sums the counter over the operator's kernels and prints bytes per operator call:

    FETCH_SIZE  is in KiB and, on gfx950, reports HALF of the bytes of a wide coalesced streaming
                read (guide: TCC_EA0_RDREQ tallied at 64 B for 128-B requests): x2.  The same factor
                is applied to every kernel here; the 4-/8-byte-per-lane column loads of kernel A are
                checked against the analytic size of the band (printed next to the counter).
    WRITE_SIZE  is in KiB, x1 (calibrated on stores of known size).

`--json` prints one JSON object {spec: corrected bytes}; `--save PREFIX` writes
PREFIX_pmc_<op>_<T>x<N>_traffic.json with the raw per-kernel counters, the corrections, the kernel
hash and (for crf) the analytic band size.  The worker process (`--worker`) is what rocprofv3 wraps.
"""
import argparse
import json
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LAUNCHES = 3


def parse_specs(text):
    out = []
    for item in text.split(","):
        name, T, N, real = item.split(":")
        out.append((name, int(T), int(N), int(real)))
    return out


def worker(specs):
    import torch
    import bench
    dev = torch.device("cuda:0")
    from taiyaki_amd import _lib
    _lib.set_strict(False)
    mark_buf = torch.zeros(2, 1024, dtype=torch.float32, device=dev)

    def mark():
        """A MARKER launch (the library's tiny device copy) cuts the kernel timeline into groups: gaps in time did,
        until a slow host under the counters left a 20 ms gap inside a group (round 5)."""
        torch.cuda.synchronize()
        _lib.check(_lib.lib().tk_devcopy_f32_dev(_lib.ptr(mark_buf[1]), _lib.ptr(mark_buf[0]), 1024, _lib.stream_ptr()), "marker")
        torch.cuda.synchronize()

    for name, T, N, real in specs:
        ops = bench.LossOps(T, N, dev, realistic_chunk_len=real or None, cat_mod=(name == "catmod"))
        fn = ops.logz_op if name == "logz" else ops.crf
        mark()
        fn()                        # warm-up group
        mark()
        for _ in range(LAUNCHES):
            fn()
        mark()
        del ops
    print("pmc-worker-done")


def read_db(db):
    con = sqlite3.connect(db)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]

    def tab(prefix):
        return [t for t in tabs if t.startswith(prefix)][0]
    kd, ks, pe = tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol"), tab("rocpd_pmc_event")
    q = ("select s.kernel_name, d.start, d.end, e.value from %s e join %s d on e.event_id = d.event_id "
         "join %s s on d.kernel_id = s.id order by d.start" % (pe, kd, ks))
    return [(r[0], r[1], r[2], float(r[3])) for r in cur.execute(q)]


def group_launches(rows, nspecs):
    """Our kernels (namespace tk) in time order, cut at the worker's marker launches: per spec one warm-up group
    and one group of LAUNCHES counted launches."""
    groups, cur = [], []
    for r in rows:
        if "devcopy_f4_kernel" in r[0]:
            if cur:
                groups.append(cur)
            cur = []
        elif "2tk" in r[0]:
            cur.append(r)
    if cur:
        groups.append(cur)
    if len(groups) != 2 * nspecs:
        raise RuntimeError("expected %d kernel groups, found %d" % (2 * nspecs, len(groups)))
    return [groups[2 * i + 1] for i in range(nspecs)]


def short(name):
    return name.replace("_ZN2tk", "").split("EvN")[0].split("EvP")[0][:48]


def band_bytes(T, N, real, R_hint=None, cat_mod=False):
    """Analytic size of what kernel A's sweeps leave for the gradient pass, per sweep: one checkpoint
    column (4 B mantissa + 2 B frame offset per cell) per live (chunk, block) pair, plus one boundary cell
    per step, 64 cells and block.  Block length as crf_band_pick_block chooses it at sharpening factor 1:
    12 steps for the plain CRF, 8 for cat-mod."""
    import numpy as np
    from taiyaki_amd import synth
    seqlens = synth.realistic_seqlens(T, N, 17001, real, 9.0) if real else synth.speedtest_seqlens(T, N)
    maxL = int(seqlens.max())
    R = 1
    while R < 4 and R * 64 * 15 < maxL:
        R *= 2
    PW = 64 * R
    KB = 8 if cat_mod else 12
    total = 0
    for L in seqlens:
        L = int(L)
        for w in range((L + PW - 1) // PW):
            a, b = w * PW, min(w * PW + PW - 1, L - 1)
            tlo, thi = max(0, a - 1), min(T - 1, b + T - L + 1)
            total += (thi // KB - tlo // KB + 1) * (PW * 6 + (PW // 64) * 4 * KB)
    return int(total)


def run_counter(counter, specs_text, outdir):
    cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "-d", outdir, "-o", "pmc", "--",
           sys.executable, os.path.abspath(__file__), "--worker", "--ops", specs_text]
    env = dict(os.environ, TMPDIR="/tmp")
    pr = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
    if pr.returncode != 0 or "pmc-worker-done" not in pr.stdout:
        raise RuntimeError("rocprofv3 %s failed: %s" % (counter, (pr.stderr or pr.stdout)[-500:]))
    dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(outdir) for f in fs if f.endswith(".db")]
    if not dbs:
        raise RuntimeError("rocprofv3 %s left no database" % counter)
    return read_db(dbs[0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ops", default="logz:4000:256:0,crf:800:128:4000")
    ap.add_argument("--worker", action="store_true")
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--save", default=None)
    args = ap.parse_args()
    specs = parse_specs(args.ops)
    if args.worker:
        return worker(specs)
    if shutil.which("rocprofv3") is None:
        raise SystemExit("rocprofv3 is not on PATH")
    import bench
    result, detail = {}, {}
    per = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        tmp = tempfile.mkdtemp(prefix="tkpmc_", dir="/tmp")
        try:
            rows = run_counter(counter, args.ops, tmp)
        finally:
            pass
        for spec, grp in zip(specs, group_launches(rows, len(specs))):
            byk = {}
            for name, _, _, val in grp:
                byk[short(name)] = byk.get(short(name), 0.0) + val / LAUNCHES
            per.setdefault(spec, {})[counter] = byk
        shutil.rmtree(tmp, ignore_errors=True)
    for spec in specs:
        name, T, N, real = spec
        fetch_kib = sum(per[spec]["FETCH_SIZE"].values())
        write_kib = sum(per[spec]["WRITE_SIZE"].values())
        fetch_b, write_b = fetch_kib * 1024 * 2.0, write_kib * 1024 * 1.0
        S = 46 if name == "catmod" else 40
        alg = 3 * T * N * S * 4
        key = "%s:%d:%d:%d" % spec
        result[key] = fetch_b + write_b
        d = dict(op=name, shape=dict(T=T, N=N, S=S, realistic_chunk_len=real),
                 command="rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace -- python tools/pmc_traffic.py "
                         "--worker --ops %s (separate passes, mean of %d launches)" % (args.ops, LAUNCHES),
                 fetch_size_kib_raw=per[spec]["FETCH_SIZE"], write_size_kib_raw=per[spec]["WRITE_SIZE"],
                 corrections="FETCH_SIZE x2 (gfx950: 128-B requests tallied at 64 B, MI355X_MICROARCH.md HBM "
                             "section), WRITE_SIZE x1; both counters sit on the memory side of L2 and include "
                             "Infinity-Cache hits",
                 fetch_bytes=fetch_b, write_bytes=write_b, traffic_bytes=fetch_b + write_b, algorithmic_bytes=alg,
                 traffic_over_algorithmic=round((fetch_b + write_b) / alg, 4), kernel_hash=bench.kernel_hash())
        if name != "logz":
            bb = band_bytes(T, N, real, cat_mod=(name == "catmod"))
            d["analytic_checkpoint_bytes_per_sweep"] = bb
            d["analytic_note"] = ("each sweep writes one checkpoint column + the boundary cells per block of the band "
                                  "(2 x %d B) and the gradient pass reads them once; scores are read by both sweeps "
                                  "and the gradient pass (3 x %d B) and the gradient is written once (%d B)"
                                  % (bb, T * N * S * 4, T * N * S * 4))
            d["analytic_traffic_bytes"] = 4 * bb + 4 * T * N * S * 4
        detail[key] = d
        if not args.json:
            print("%-22s fetch %8.1f MB (raw %8.1f MiB x2)  write %8.1f MB  total %8.1f MB = %.2fx algorithmic %s"
                  % (key, fetch_b / 1e6, fetch_kib / 1024, write_b / 1e6, (fetch_b + write_b) / 1e6,
                     (fetch_b + write_b) / alg,
                     ("(analytic %.1f MB)" % (d["analytic_traffic_bytes"] / 1e6)) if name != "logz" else ""))
            for k, v in sorted(per[spec]["FETCH_SIZE"].items()):
                print("      %-50s fetch %10.1f KiB   write %10.1f KiB" % (k, v, per[spec]["WRITE_SIZE"].get(k, 0.0)))
        if args.save:
            path = "%s_pmc_%s_%dx%d%s_traffic.json" % (args.save, name, T, N, "r" if real else "")
            with open(path, "w") as fh:
                json.dump(d, fh, indent=1)
    if args.json:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
