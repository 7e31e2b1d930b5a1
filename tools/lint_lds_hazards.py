#!/usr/bin/env python3
"""Static check of the device code for reads of LDS-load destinations that are still in flight.

Some kernels (csrc/cm_critic_fused.h, csrc/cm_gru_step2.h) issue their LDS reads through inline asm and wait for them by hand
(cm_common.h: cf_lds128 / cf_wait) so that the reads run ahead of the MFMAs that consume them.  The compiler does not know that such a
read is asynchronous: if it ever copies or otherwise reads the destination registers between the read and its s_waitcnt (a phi copy at a
branch did exactly that once), the kernel computes garbage without any diagnostic.  This linter walks the compiler's assembly
(`-save-temps=obj` of the library build: cleanmarl_amd/build/libcleanmarl_hip.so.obj/*-gfx950.s; `python -m cleanmarl_amd.build --lint`) kernel by kernel with the
in-order LGKM queue: reads between `;;#ASMSTART` and `;;#ASMEND` are the hand-issued ones; an instruction that reads or overwrites
one of their destination registers ahead of the covering `s_waitcnt lgkmcnt(N)`, or a branch / label while one is pending (the helpers
are for straight-line pipelines), is reported.

    python tools/lint_lds_hazards.py [file.s ...]      exit code 1 if a hazard is found
"""
import os
import re
import sys

REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")
LGKM_OTHER = ("s_load", "s_buffer_load", "s_memtime", "s_memrealtime", "s_dcache", "s_sendmsg", "ds_")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def lint_kernel(name, lines):
    """lines: (in_asm, text) per instruction / label of one kernel."""
    pending, problems = [], []   # FIFO of (destination registers, hand-issued?) per LGKM operation
    for in_asm, ins in lines:
        if ins.endswith(":"):  # label
            if any(h for _, h in pending):
                problems.append(f"{name}: label {ins} reached with a hand-issued LDS read in flight")
            continue
        op, _, rest = ins.partition(" ")
        ops = [o.strip() for o in rest.split(",")] if rest else []
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", rest)
            if m:
                while len(pending) > int(m.group(1)):
                    pending.pop(0)
            continue
        hand = set().union(*[d for d, h in pending if h]) if pending else set()
        if op.startswith("s_cbranch") or op == "s_branch" or op == "s_endpgm" or op.startswith("s_setpc"):
            if hand:
                problems.append(f"{name}: `{ins}` with a hand-issued LDS read in flight")
            continue
        is_load = op.startswith(("ds_read", "ds_bpermute", "ds_permute", "ds_swizzle"))
        if hand:
            if op.startswith(("ds_write", "global_store", "scratch_store", "buffer_store")):
                touched = regs(rest)
            else:
                touched = regs(",".join(ops[1:])) if len(ops) > 1 else set()
                if not is_load and ops:
                    touched |= regs(ops[0])
            hit = touched & hand
            if hit:
                problems.append(f"{name}: `{ins}` touches {sorted(hit)[:4]} while hand-issued LDS reads are in flight")
        if is_load:
            pending.append((regs(ops[0]) if ops else set(), in_asm))
        elif op.startswith(LGKM_OTHER):
            pending.append((set(), False))
    return problems


def lint_file(path):
    problems, nk, nhand = [], 0, 0
    name, body, in_asm = None, [], False
    for ln in open(path):
        t = ln.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if t.startswith(".amdhsa_kernel") or t.startswith(".section") or t.startswith(".end_amdhsa_kernel"):
            continue
        m = re.match(r"^(_Z\w+|[A-Za-z_]\w*):\s*(;.*)?$", ln)
        if m and not ln.startswith("."):
            if name and body:
                nk += 1
                problems += lint_kernel(name, body)
            name, body = m.group(1), []
            continue
        if name is None or not t or t.startswith(";") or t.startswith("//"):
            continue
        t = t.split(";")[0].strip()
        if not t:
            continue
        if t.startswith(".") and not t.endswith(":"):
            continue  # directive
        if in_asm and t.startswith("ds_read"):
            nhand += 1
        body.append((in_asm, t))
        if t == "s_endpgm":
            nk += 1
            problems += lint_kernel(name, body)
            name, body = None, []
    return problems, nk, nhand


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(here, "cleanmarl_amd", "build", "libcleanmarl_hip.so.obj")
    files = sys.argv[1:] or sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith("gfx950.s"))
    total, bad = 0, 0
    for f in files:
        probs, nk, nhand = lint_file(f)
        for p in probs[:20]:
            print(p)
        print(f"{os.path.basename(f)}: {nk} kernels, {nhand} hand-issued LDS reads, {len(probs)} hazards")
        total += nhand
        bad += len(probs)
    sys.exit(1 if bad else 0)
