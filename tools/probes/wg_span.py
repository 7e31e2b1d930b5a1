#!/usr/bin/env python3
"""Per-workgroup timing of the actor pass (cm_clock_probe): when every workgroup of one launch entered / left its tile loop, on which XCD /
CU it ran and at which shader clock.  usage: python tools/probes/wg_span.py [envs=4096] [agents=8] [steps=128]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from cleanmarl_amd import _native as N  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
w = bench.Workload("cfg3", E, 0, dev)
L = w.learner
for _ in range(3):
    w.one_step()
torch.cuda.synchronize()
clk = torch.zeros(512, 4, dtype=torch.int64, device=dev)
N.check(N.load().cm_clock_probe(N.ptr(clk)), "probe")
b = w.roll.collect(L.actor, w.aspec)
L.compute_targets(b)
L.wait_critic()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
L.actor_pass(b, N.stream_ptr())
e1.record()
torch.cuda.synchronize()
N.load().cm_clock_probe(None)
c = clk.cpu()
c = c[c[:, 3] > 0]
t0 = int(c[:, 2].min())
st, en = (c[:, 2] - t0).double() / 1e5, (c[:, 3] - t0).double() / 1e5
busy = en - st
ghz = c[:, 0].double() / (busy * 1e-3) / 1e9
xcc = (c[:, 1] >> 32) & 0xF
hw = c[:, 1] & 0xFFFFFFFF
cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 0x1, (hw >> 13) & 0x7
print(f"launch {e0.elapsed_time(e1):.4f} ms by HIP events (pass + reduction); {c.shape[0]} workgroups")
print(f"entry: last {float(st.max()):.4f} ms; exit: first {float(en.min()):.4f} median {float(en.median()):.4f} last {float(en.max()):.4f}; busy mean {float(busy.mean()):.4f}")
qs = torch.tensor([0.0, 0.1, 0.25, 0.5, 0.75, 0.9, 0.95, 0.99, 1.0], dtype=torch.float64)
print("busy quantiles (ms):", [round(float(v), 4) for v in torch.quantile(busy, qs)])
print("clock quantiles (GHz):", [round(float(v), 3) for v in torch.quantile(ghz, qs)])
for x in range(8):
    m = xcc == x
    if m.any():
        print(f"XCD {x}: {int(m.sum())} workgroups, busy mean {float(busy[m].mean()):.4f} max {float(busy[m].max()):.4f}, clock mean {float(ghz[m].mean()):.3f}")
slow = torch.argsort(busy, descending=True)[:24]
print("slowest 24: (wg, xcd, se, sh, cu, busy ms, GHz)")
for i in slow.tolist():
    print("  ", i, int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i]), round(float(busy[i]), 4), round(float(ghz[i]), 3))
# workgroups per (xcd, se, sh, cu): more than 2 = a CU holding three workgroups' worth of time?
key = (xcc * 4096 + se * 256 + sh * 16 + cu).tolist()
from collections import Counter
cnt = Counter(key)
print("workgroups per CU:", Counter(cnt.values()), "distinct CUs:", len(cnt))
by = {}
for k, bz in zip(key, busy.tolist()):
    by.setdefault(k, []).append(bz)
solo = [v[0] for v in by.values() if len(v) == 1]
pair = [x for v in by.values() if len(v) == 2 for x in v]
more = [x for v in by.values() if len(v) > 2 for x in v]
for nm, v in (("1 per CU", solo), ("2 per CU", pair), (">2 per CU", more)):
    if v:
        print(f"{nm}: {len(v)} workgroups, busy mean {sum(v) / len(v):.4f} max {max(v):.4f}")
