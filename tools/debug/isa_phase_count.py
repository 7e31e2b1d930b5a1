#!/usr/bin/env python3
"""Static instruction mix of one kernel of the library build's device assembly, split at s_barrier.

    python tools/debug/isa_phase_count.py <file.s> <mangled-name-substring> [--loop]

Prints per barrier-delimited segment: MFMA (by shape), other VALU (by mnemonic class), LDS reads / writes, global, scalar; with --loop only
the instructions between the largest backward branch's target and the branch (the persistent tile loop).  Straight-line count, not
weighted by trip counts of inner loops (the tile loop of k_mlp is fully unrolled inside).
"""
import collections
import re
import sys


def kernel_body(path, pat):
    body, on = [], False
    for ln in open(path):
        t = ln.split(";")[0].strip() if not ln.lstrip().startswith(";") else ln.strip()
        if not on:
            if t.endswith(":") and pat in t and not t.startswith("."):
                on = True
            continue
        if t.startswith("s_endpgm"):
            body.append(t)
            break
        if not t or t.startswith((";", ".")) and not t.endswith(":"):
            continue
        body.append(t.split(";")[0].strip())
    return [b for b in body if b]


def klass(op):
    if op.startswith("v_mfma"):
        return "mfma:" + op.split("_f32_")[1] if "_f32_" in op else "mfma"
    if op.startswith("v_accvgpr"):
        return "v_accvgpr"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_read") or op.startswith("ds_bpermute") or op.startswith("ds_swizzle"):
        return "ds_read"
    if op.startswith("ds_"):
        return "ds_write"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    body = kernel_body(path, pat)
    labels = {b[:-1]: i for i, b in enumerate(body) if b.endswith(":")}
    lo, hi = 0, len(body)
    if "--loop" in sys.argv:
        best = (0, 0, 0)
        for i, b in enumerate(body):
            m = re.match(r"s_cbranch_\w+\s+(\S+)|s_branch\s+(\S+)", b)
            if m:
                tgt = labels.get(m.group(1) or m.group(2))
                if tgt is not None and tgt < i and i - tgt > best[0]:
                    best = (i - tgt, tgt, i)
        _, lo, hi = best
    seg, segs = collections.Counter(), []
    detail = collections.Counter()
    for b in body[lo:hi]:
        if b.endswith(":"):
            continue
        op = b.split()[0]
        if op == "s_barrier":
            segs.append((seg, detail))
            seg, detail = collections.Counter(), collections.Counter()
            continue
        k = klass(op)
        seg[k] += 1
        if k == "valu":
            detail[re.sub(r"_e32|_e64|_dpp|_sdwa", "", op)] += 1
    segs.append((seg, detail))
    tot, totd = collections.Counter(), collections.Counter()
    for i, (s, d) in enumerate(segs):
        tot.update(s)
        totd.update(d)
        mf = " ".join(f"{k[5:]}={v}" for k, v in sorted(s.items()) if k.startswith("mfma"))
        print(f"seg {i:2d}: valu {s['valu']:4d} acc {s['v_accvgpr']:3d} dsr {s['ds_read']:3d} dsw {s['ds_write']:3d} vmem {s['vmem']:3d} "
              f"salu {s['salu']:4d} wait {s['s_waitcnt']:3d} nop {s['s_nop']:3d} | {mf} | " + " ".join(f"{k[2:]}:{v}" for k, v in d.most_common(6)))
    print("total:", dict(tot))
    print("valu by mnemonic:", " ".join(f"{k}:{v}" for k, v in totd.most_common(40)))


if __name__ == "__main__":
    main()
