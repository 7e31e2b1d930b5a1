#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes (separate --pmc runs, as MI355X_MICROARCH.md prescribes) per kernel.
usage: python tools/pmc_summary.py <dir with pmc_fetch/ pmc_write/ pmc_mfma/> > profiles/rNN_pmc_summary.txt
Writes <dir>/pmc_summary.json as well; with `--emit <path> <source text>` also the record bench.py reads for roofline.traffic
(profiles/pmc_dominant_kernel.json): the actor pass's HBM bytes / MFMA-busy fraction stamped with the hash of the library's sources.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of
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

if "--emit" in sys.argv:
    at = sys.argv.index("--emit")
    path, source = sys.argv[at + 1], sys.argv[at + 2]
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from cleanmarl_amd.build import source_hash
    actor = [k for k in out if "k_mlp<1, 2," in k]
    if not actor:
        raise SystemExit("pmc_summary --emit: no k_mlp<1, 2, ...> (actor pass of config 3) among the profiled kernels")
    k = max(actor, key=lambda k: out[k]["hbm_bytes"])
    pick = lambda pat: next((dict(hbm_bytes_per_launch=out[q]["hbm_bytes"], mfma_busy_frac=out[q]["mfma_busy_frac"]) for q in out if pat in q), None)
    rec = dict(kernel=k, hbm_bytes_per_launch=out[k]["hbm_bytes"], mfma_busy_frac=out[k]["mfma_busy_frac"], source=source, workload="cfg3",
               source_hash=source_hash(),
               other_kernels={n: v for n, v in (("k_critic_fused<6>", pick("k_critic_fused<6>")), ("k_mlp<0,0> value pass", pick("k_mlp<0, 0,")),
                                                ("k_rollout_spread64s", pick("k_rollout_spread64s"))) if v})
    json.dump(rec, open(path, "w"), indent=1)
    print(f"wrote {path}: {k[:60]} {rec['hbm_bytes_per_launch'] / 1e6:.1f} MB per launch, sources {rec['source_hash']}")
