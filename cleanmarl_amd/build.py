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


ASM_DIR = os.path.join(HERE, "build", "asm")
HAND_PIPELINED = ("cm_mlp_critic.hip", "cm_gru.hip")  # translation units that use cm_common.h's cf_lds128 / cf_wait


def emit_asm(force=False):
    """Device assembly (hipcc -S --cuda-device-only) of the translation units with hand-issued LDS reads, for
    tools/lint_lds_hazards.py.  Returns the .s paths; up-to-date files are kept."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(ASM_DIR, exist_ok=True)
    deps = glob.glob(os.path.join(CSRC, "*.h"))

    def one(name):
        src, out = os.path.join(CSRC, name), os.path.join(ASM_DIR, name + ".s")
        if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps + [src]):
            return out
        cmd = [HIPCC] + [f for f in FLAGS if f not in ("-shared", "-fPIC")] + ["-S", "--cuda-device-only", src, "-o", out + ".tmp"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc -S failed on {src}:\n" + r.stdout + r.stderr)
        os.replace(out + ".tmp", out)
        return out
    with ThreadPoolExecutor(max_workers=len(HAND_PIPELINED)) as ex:
        return list(ex.map(one, HAND_PIPELINED))


def lint_hand_pipelines(force=False):
    """emit_asm + the in-flight-register check; raises on a hazard.  Returns (hand-issued reads, kernels) checked."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("lint_lds_hazards", os.path.join(HERE, "..", "tools", "lint_lds_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    reads = kernels = 0
    for f in emit_asm(force):
        probs, nk, nhand = mod.lint_file(f)
        if probs:
            raise RuntimeError(f"{os.path.basename(f)}: registers of in-flight LDS reads are touched before their s_waitcnt:\n  " + "\n  ".join(probs[:10]))
        reads, kernels = reads + nhand, kernels + nk
    if reads == 0:
        raise RuntimeError("lint_hand_pipelines: no hand-issued LDS read found -- the asm markers changed?")
    return reads, kernels


if __name__ == "__main__":
    import sys
    if "--asm" in sys.argv:
        print("hand-issued LDS reads / kernels checked:", lint_hand_pipelines(force=True))
    elif "--prof" in sys.argv:  # phase-profiling build used by tools/phase_prof.py
        print(build_native(force=True, out=os.path.join(HERE, "libcleanmarl_hip_prof.so"), extra_flags=["-DCM_PHASE_PROF"]))
    else:
        print(build_native(force=True, verbose="-v" in sys.argv))
