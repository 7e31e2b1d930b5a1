#!/usr/bin/env python3
"""A/B library for same-box comparisons: compiles the csrc/ of a git revision (default HEAD) into cleanmarl_amd/libcleanmarl_hip_ab.so, which
bench.py / the tools load through CM_LIB_PATH (tools/gpu/run.sh "ab CM_LIB_PATH <main .so> <ab .so> -- ..." runs the working tree's build and this one alternately)."""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rev = sys.argv[1] if len(sys.argv) > 1 else "HEAD"
flags = [f for f in sys.argv[2:] if not f.startswith("--out=")]
suffix = ([f[6:] for f in sys.argv[2:] if f.startswith("--out=")] or ["ab"])[0]   # library name: libcleanmarl_hip_<suffix>.so
TMP = f"/tmp/ab_src_{suffix}"
if rev == "WORK":  # the working tree's sources (with the given -D flags) instead of a revision
    subprocess.check_call(f"rm -rf {TMP} && mkdir -p {TMP}/cleanmarl_amd && cp -r {ROOT}/cleanmarl_amd/csrc {TMP}/cleanmarl_amd/ && cp -r {ROOT}/include {TMP}/", shell=True)
else:
    subprocess.check_call(f"rm -rf {TMP} && mkdir -p {TMP} && git -C {ROOT} archive {rev} cleanmarl_amd/csrc include | tar -x -C {TMP}", shell=True)
srcs = sorted(glob.glob(f"{TMP}/cleanmarl_amd/csrc/*.hip"))


def comp(s):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed"] + flags + ["-c", s, "-o", s + ".o"])
    return s + ".o"


with ThreadPoolExecutor(8) as ex:
    objs = list(ex.map(comp, srcs))
out = os.path.join(ROOT, "cleanmarl_amd", f"libcleanmarl_hip_{suffix}.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
print("built", out, "from", rev, flags)
