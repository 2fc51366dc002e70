#!/usr/bin/env python3
"""Builds the committed rocprofv3 summaries of a round from the raw output under gpurun_out/ (scratch).

usage: python profiles/make_summary.py <tag>      e.g. r01_final_flipout_conv_s10
  gpurun_out/prof_stats/*/*_kernel_stats.csv          -> profiles/<tag>_kernel_stats.csv   (copied)
  gpurun_out/prof_fetch, prof_write/*/*_counter_collection.csv (separate --pmc FETCH_SIZE / WRITE_SIZE passes)
                                                      -> profiles/<tag>_pmc_summary.csv: per kernel symbol the mean
     FETCH_SIZE / WRITE_SIZE per launch (KB, as rocprofv3 reports them) and (2*FETCH + WRITE)/1024 MB, i.e. with the
     gfx950 read correction of MI355X_MICROARCH.md (FETCH_SIZE counts a 128-B read request as 64 B).
"""
import collections
import csv
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
out = os.path.join(ROOT, "profiles")


def per_kernel(dirname, counter):
    fs = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", dirname, "*", "*_counter_collection.csv")), key=os.path.getmtime)[-1:]
    acc, cnt = collections.defaultdict(float), collections.Counter()
    for f in fs:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            acc[r["Kernel_Name"]] += float(r["Counter_Value"])
            cnt[r["Kernel_Name"]] += 1
    return {k: (acc[k] / cnt[k], cnt[k]) for k in acc}


st = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "prof_stats", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
if st:
    shutil.copy(st[-1], os.path.join(out, f"{tag}_kernel_stats.csv"))
fe, wr = per_kernel("prof_fetch", "FETCH_SIZE"), per_kernel("prof_write", "WRITE_SIZE")
rows = []
for k in sorted(set(fe) | set(wr), key=lambda k: -(2 * fe.get(k, (0, 0))[0] + wr.get(k, (0, 0))[0]) * max(fe.get(k, (0, 1))[1], 1)):
    f, n = fe.get(k, (0.0, 0))
    w, _ = wr.get(k, (0.0, 0))
    rows.append([k, n, round(f, 1), round(w, 1), round((2 * f + w) / 1024, 1)])
with open(os.path.join(out, f"{tag}_pmc_summary.csv"), "w", newline="") as fh:
    wtr = csv.writer(fh)
    wtr.writerow(["kernel", "launches", "FETCH_SIZE_KB_per_launch", "WRITE_SIZE_KB_per_launch",
                  "hbm_MB_per_launch_(2*fetch+write)"])
    wtr.writerows(rows)
print("wrote", len(rows), "kernels")
