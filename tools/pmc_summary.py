#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes (separate --pmc runs, as MI355X_MICROARCH.md prescribes) per kernel.
usage: python tools/pmc_summary.py <dir with pmc_fetch/ pmc_write/ pmc_mfma/> > profiles/rNN_pmc_summary.txt
Writes <dir>/pmc_summary.json as well.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of
the bytes of a wide (16 B/lane) coalesced streaming read, so the read side is doubled (guide, §HBM)."""
import collections
import csv
import glob
import json
import os
import sys

root = sys.argv[1]


def agg(sub, counter):
    d = collections.defaultdict(list)
    for path in glob.glob(os.path.join(root, sub, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in d.items()}


fetch, write = agg("pmc_fetch", "FETCH_SIZE"), agg("pmc_write", "WRITE_SIZE")
mfma, gui = agg("pmc_mfma", "SQ_VALU_MFMA_BUSY_CYCLES"), agg("pmc_mfma", "GRBM_GUI_ACTIVE")
out = {}
print(f"{'kernel':72s} {'fetch_MB(x2)':>12s} {'write_MB':>9s} {'hbm_MB':>9s} {'mfma_busy':>9s}")
for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, 0) + write.get(k, 0))):
    if "at::native" in k or "rocclr" in k:
        continue
    f = 2.0 * fetch.get(k, 0.0) * 1024
    w = write.get(k, 0.0) * 1024
    util = None
    if k in mfma and gui.get(k):
        util = (mfma[k] / (256 * 4)) / (gui[k] / 8)  # busy cycles per SIMD / active cycles per XCD
    out[k] = dict(fetch_bytes=f, write_bytes=w, hbm_bytes=f + w, mfma_busy_frac=util)
    print(f"{k[:72]:72s} {f/1e6:12.1f} {w/1e6:9.1f} {(f+w)/1e6:9.1f} {'' if util is None else f'{100*util:8.1f}%'}")
json.dump(out, open(os.path.join(root, "pmc_summary.json"), "w"), indent=1)
