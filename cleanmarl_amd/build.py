"""In-tree build of libcleanmarl_hip.so (hipcc, gfx950 only).  hipcc cross-compiles without a GPU."""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcleanmarl_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-pass-failed"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build_native(force=False, verbose=False):
    """Compile every csrc/*.hip into one shared library next to this file."""
    if not force and not is_stale():
        return LIB
    cmd = [HIPCC] + FLAGS + ["-o", LIB + ".tmp"] + sources()
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build_native(force=True, verbose=True))
