"""In-tree build of libcleanmarl_hip.so (hipcc, gfx950 only).  hipcc cross-compiles without a GPU."""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcleanmarl_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-pass-failed"]
FILE_FLAGS = {}  # per-file extra flags (none: the wave-private actor probe that needed some lives in tools/probes/actor16/)


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(args):
    src, obj, extra, verbose = args
    cmd = [HIPCC] + [f for f in FLAGS if f != "-shared"] + extra + FILE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n" + r.stdout + r.stderr)
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
    os.replace(out + ".tmp", out)
    return out


if __name__ == "__main__":
    import sys
    if "--prof" in sys.argv:  # phase-profiling build used by tools/phase_prof.py
        print(build_native(force=True, out=os.path.join(HERE, "libcleanmarl_hip_prof.so"), extra_flags=["-DCM_PHASE_PROF"]))
    else:
        print(build_native(force=True, verbose="-v" in sys.argv))
