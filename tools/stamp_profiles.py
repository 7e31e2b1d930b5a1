#!/usr/bin/env python3
"""Copy the artefacts of tools/refresh_profiles.sh from gpurun_out/<tag>/ to profiles/<tag>_* and write the stamp of the run -- commit and
source hash of the library they were taken on (gpurun_out/<tag>/MANIFEST.txt) -- INTO each of them: JSON records get a "stamp" key, text
files a first comment line; CSV files stay byte-identical (rocprofv3's format) and are covered by profiles/<tag>_MANIFEST.txt, which lists
every file with its size.  Also installs pmc_dominant_kernel.json (the record bench.py reads for roofline.traffic).
    python tools/stamp_profiles.py r06"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
manifest = open(os.path.join(src, "MANIFEST.txt")).read()
stamp = manifest.splitlines()[0].strip()
rename = {"bench.json": "bench.json", "gputests.txt": "gputests_log.txt", "parity_observed.txt": "gputests.txt"}
skip = {"bench.err", "coma.err", "coma128.err", "pmc_dominant_kernel.json", "issue_counters.json", "MANIFEST.txt"}  # the two .json records are installed below
copied = []
for f in sorted(os.listdir(src)):
    p = os.path.join(src, f)
    if not os.path.isfile(p) or f in skip or f.startswith("quick_") or os.path.getsize(p) == 0:
        continue
    out = os.path.join(dst, f"{tag}_{rename.get(f, f)}")
    if f.endswith(".json"):
        lines = [ln for ln in open(p) if ln.startswith("{")]
        if not lines:
            continue
        rec = json.loads(lines[-1])
        rec["stamp"] = stamp
        json.dump(rec, open(out, "w"))
        open(out, "a").write("\n")
    elif f.endswith(".csv"):
        shutil.copyfile(p, out)
    else:
        body = open(p, errors="replace").read()
        open(out, "w").write(f"# {stamp}\n" + body)
    copied.append(os.path.basename(out))
open(os.path.join(dst, f"{tag}_MANIFEST.txt"), "w").write(manifest + "\ncopied to profiles/:\n" + "\n".join("  " + c for c in copied) + "\n")
pmc = os.path.join(src, "pmc_dominant_kernel.json")
if os.path.exists(pmc):
    shutil.copyfile(pmc, os.path.join(dst, "pmc_dominant_kernel.json"))
    shutil.copyfile(pmc, os.path.join(dst, f"{tag}_pmc_dominant_kernel.json"))
issue = os.path.join(src, "issue_counters.json")  # tools/gpu/r06_issue_pmc.sh: the record bench.py reads for roofline.issue (carries its own source_hash)
if os.path.exists(issue):
    shutil.copyfile(issue, os.path.join(dst, "issue_counters.json"))
print(stamp)
print(f"{len(copied)} files -> profiles/{tag}_*")
