"""In-tree build of libcleanmarl_hip.so (hipcc, gfx950 only).  hipcc cross-compiles without a GPU."""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcleanmarl_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-pass-failed"]
INCREMENTAL = False  # set by `python -m cleanmarl_amd.build -i` (development loop); __graft_entry__.build() always compiles everything
FILE_FLAGS = {}  # per-file extra flags (none: the wave-private actor probe that needed some lives in tools/probes/actor16/)


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def source_hash():
    """sha256 over every source the library is compiled from (csrc/*.hip, csrc/*.h, include/*.h; names + contents, sorted): stamps
    measurements that cannot be repeated inside bench.py (the PMC passes behind profiles/pmc_dominant_kernel.json) with the kernels they
    were taken on -- bench.py drops a `roofline.traffic` whose stamp is not the hash of the sources of the library it is timing."""
    import hashlib
    h = hashlib.sha256()
    for f in sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + sorted(glob.glob(os.path.join(HERE, "..", "include", "*.h"))):
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def _fresh(obj, flags):
    """True when `obj` was compiled with `flags` and is newer than its source and every header the compiler listed for it (-MMD)."""
    dep, stamp = obj + ".d", obj + ".flags"
    if not (os.path.exists(obj) and os.path.exists(dep) and os.path.exists(stamp)) or open(stamp).read() != " ".join(flags):
        return False
    deps = open(dep).read().replace("\\\n", " ").split(":", 1)[-1].split()
    t = os.path.getmtime(obj)
    return all(os.path.exists(d) and os.path.getmtime(d) <= t for d in deps)


def _compile(args):
    src, obj, extra, verbose = args
    # -save-temps=obj: the device assembly (<stem>-hip-amdgcn-amd-amdhsa-gfx950.s) lands next to the object for lint_hand_pipelines()
    flags = [f for f in FLAGS if f != "-shared"] + ["-save-temps=obj"] + extra + FILE_FLAGS.get(os.path.basename(src), [])
    if INCREMENTAL and _fresh(obj, flags):  # python -m cleanmarl_amd.build -i: only the translation units whose sources changed
        return obj
    cmd = [HIPCC] + flags + ["-MMD", "-MF", obj + ".d", "-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n" + r.stdout + r.stderr)
    open(obj + ".flags", "w").write(" ".join(flags))
    stem = os.path.join(os.path.dirname(obj), os.path.splitext(os.path.basename(src))[0])
    for junk in glob.glob(stem + "-*.hipi") + glob.glob(stem + "-*.bc") + glob.glob(stem + "-host-*.s") + glob.glob(stem + "-*.out*"):
        os.remove(junk)
    return obj


def build_native(force=False, verbose=False, out=None, extra_flags=()):
    """Compile every csrc/*.hip (one translation unit per file, in parallel) and link them into ONE shared
    library next to this file.  hipcc cross-compiles for gfx950 without a GPU present."""
    out = out or LIB
    if not force and out == LIB and not is_stale():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(HERE, "build", os.path.basename(out) + ".obj")
    os.makedirs(objdir, exist_ok=True)
    jobs = [(s, os.path.join(objdir, os.path.basename(s) + ".o"), list(extra_flags), verbose) for s in sources()]
    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(_compile, jobs))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out + ".tmp"] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
    if out == LIB:  # the hand-issued LDS pipelines depend on what THIS compiler did around them: a hazard fails the build --
        try:        # BEFORE the library is installed, so that a hazardous build never becomes the non-stale library the next import loads
            _lint(device_asm(out))
        except Exception:
            os.remove(out + ".tmp")
            raise
    os.replace(out + ".tmp", out)
    return out


def device_asm(lib=None):
    """The device assembly files written next to the objects of the last build of `lib` (build_native compiles with -save-temps=obj)."""
    objdir = os.path.join(HERE, "build", os.path.basename(lib or LIB) + ".obj")
    return sorted(glob.glob(os.path.join(objdir, "*-hip-amdgcn-amd-amdhsa-gfx950.s")))


def lint_hand_pipelines():
    """tools/lint_lds_hazards.py over the assembly of every translation unit of the in-tree library (builds it first if its assembly is
    missing or stale); raises on a hazard.  Returns (hand-issued LDS reads, kernels) checked."""
    files = device_asm()
    by_stem = {os.path.basename(f).split("-hip-amdgcn")[0]: f for f in files}
    stale = is_stale() or any(os.path.splitext(os.path.basename(src))[0] not in by_stem or
                              os.path.getmtime(by_stem[os.path.splitext(os.path.basename(src))[0]]) < os.path.getmtime(src) for src in sources())
    if stale:
        build_native(force=True)  # lints as its last step
        files = device_asm()
    return _lint(files)


def _lint(files):
    import importlib.util
    script = os.path.join(HERE, "..", "tools", "lint_lds_hazards.py")
    if not os.path.exists(script):  # a package copied without tools/: the check is a build-tree facility, its absence is not a build error
        import warnings
        warnings.warn("tools/lint_lds_hazards.py not found: hand-issued LDS pipelines NOT checked for this build")
        return 0, 0
    spec = importlib.util.spec_from_file_location("lint_lds_hazards", script)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    reads = kernels = 0
    for f in files:
        probs, nk, nhand = mod.lint_file(f)
        if probs:
            raise RuntimeError(f"{os.path.basename(f)}: registers of in-flight LDS reads are touched before their s_waitcnt:\n  " + "\n  ".join(probs[:10]))
        reads, kernels = reads + nhand, kernels + nk
    if reads == 0:
        raise RuntimeError("lint_hand_pipelines: no hand-issued LDS read found -- the asm markers changed?")
    return reads, kernels


if __name__ == "__main__":
    import sys
    if "--lint" in sys.argv:
        print("hand-issued LDS reads / kernels checked:", lint_hand_pipelines())
    elif "--prof" in sys.argv:  # phase-profiling build used by tools/phase_prof.py
        print(build_native(force=True, out=os.path.join(HERE, "libcleanmarl_hip_prof.so"), extra_flags=["-DCM_PHASE_PROF"]))
    else:
        INCREMENTAL = "-i" in sys.argv
        print(build_native(force=True, verbose="-v" in sys.argv))
