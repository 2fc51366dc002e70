#!/usr/bin/env python3
"""Builds the committed rocprofv3 summaries of a round from the raw output under gpurun_out/ (scratch; written by
tests/prof_gpu.sh on the MI355X box).

usage: python profiles/make_summary.py <tag> [src]     e.g. r02_flipout_conv_s10 [gpurun_out/keep_flipout_conv_s10]
  (src defaults to gpurun_out: the raw output of the last tests/prof_gpu.sh; tests/round_gpu.sh keeps one copy per workload)
  gpurun_out/prof_stats/*/*_kernel_stats.csv          -> profiles/<tag>_kernel_stats.csv   (copied)
  gpurun_out/prof_fetch, prof_write/*/*_counter_collection.csv (separate --pmc FETCH_SIZE / WRITE_SIZE passes)
      -> profiles/<tag>_pmc_summary.csv: per kernel symbol the mean FETCH_SIZE / WRITE_SIZE per launch (KB, as rocprofv3
         reports them), launches per step, and (2*FETCH + WRITE)/1024 MB per launch, i.e. with the gfx950 read correction
         of MI355X_MICROARCH.md (FETCH_SIZE counts a 128-B read request as 64 B)
  gpurun_out/pmc_sq, pmc_sq2/*/*_counter_collection.csv (two SQ passes)
      -> profiles/<tag>_sq_summary.csv: per kernel symbol the mean of every SQ counter per launch and
         mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * launch duration * clock) is left to the reader: the row
         carries the raw counters plus valu_per_mfma and lds_conflict_share
"""
import collections
import csv
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
SRC = os.path.join(ROOT, sys.argv[2]) if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out")
out = os.path.join(ROOT, "profiles")


def newest(dirname):
    fs = sorted(glob.glob(os.path.join(SRC, dirname, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
    return fs[-1:] if fs else []


def per_kernel(dirname):
    """{kernel: {counter: (mean per launch, launches)}}"""
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(collections.Counter)
    for f in newest(dirname):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[r["Kernel_Name"]][r["Counter_Name"]] += 1
    return {k: {c: (v / cnt[k][c], cnt[k][c]) for c, v in d.items()} for k, d in acc.items()}


def steps_of(logname):
    """warm-up + timed + per-kernel pass steps of the profiled bench command (for launches per step)"""
    try:
        for line in open(os.path.join(SRC, logname)):
            if line.startswith("{"):
                import json
                d = json.loads(line)
                return d["steps"] + d["warmup"] + min(d["steps"], 10)
    except OSError:
        pass
    return None


st = sorted(glob.glob(os.path.join(SRC, "prof_stats", "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)
if st:
    shutil.copy(st[-1], os.path.join(out, f"{tag}_kernel_stats.csv"))
fe, wr = per_kernel("prof_fetch"), per_kernel("prof_write")
nsteps = steps_of("prof_fetch.log")
rows = []
for k in set(fe) | set(wr):
    f, n = fe.get(k, {}).get("FETCH_SIZE", (0.0, 0))
    w, _ = wr.get(k, {}).get("WRITE_SIZE", (0.0, 0))
    rows.append([k, n, round(n / nsteps, 2) if nsteps else "", round(f, 1), round(w, 1), round((2 * f + w) / 1024, 2)])
rows.sort(key=lambda r: -(r[5] * (r[2] or 1)))
with open(os.path.join(out, f"{tag}_pmc_summary.csv"), "w", newline="") as fh:
    wtr = csv.writer(fh)
    wtr.writerow(["kernel", "launches", "launches_per_step", "FETCH_SIZE_KB_per_launch", "WRITE_SIZE_KB_per_launch",
                  "hbm_MB_per_launch_(2*fetch+write)"])
    wtr.writerows(rows)
sq = per_kernel("pmc_sq")
for k, d in per_kernel("pmc_sq2").items():
    sq.setdefault(k, {}).update(d)
names = sorted({c for d in sq.values() for c in d})
with open(os.path.join(out, f"{tag}_sq_summary.csv"), "w", newline="") as fh:
    wtr = csv.writer(fh)
    wtr.writerow(["kernel", "launches"] + names + ["valu_per_mfma", "lds_conflict_share", "wait_any_share"])
    for k in sorted(sq, key=lambda k: -sq[k].get("SQ_WAVE_CYCLES", (0, 0))[0] * sq[k].get("SQ_WAVE_CYCLES", (0, 1))[1]):
        d = sq[k]
        g = lambda c: d.get(c, (0.0, 0))[0]
        wtr.writerow([k, d.get("SQ_WAVE_CYCLES", (0, 0))[1]] + [round(g(c)) for c in names] +
                     [round(g("SQ_INSTS_VALU") / g("SQ_INSTS_MFMA"), 2) if g("SQ_INSTS_MFMA") else "",
                      round(g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"), 3) if g("SQ_LDS_IDX_ACTIVE") else "",
                      round(g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), 3) if g("SQ_WAVE_CYCLES") else ""])
print("wrote", len(rows), "kernels (traffic),", len(sq), "kernels (SQ)")
