"""Host environments -> device rollout buffer without host-side collation (second half).

The reference exchanges one pickled message per env and step and collates Python lists into padded tensors at the end of the
episode (cleanmarl/mappo_multienvs.py:393-453, 109-157); `driver.host_rollout_shm` already batches the exchange through shared
memory but still stages every step through pageable numpy copies (obs -> torch -> .to(device), actions .cpu()) and builds the batch
on the host.  Here the shared blocks the env workers write (obs, state, availability) and read (actions) are page-locked ONCE
(hipHostRegister through torch's cudart binding), and a step is

    3 async H2D copies  shm -> rollout buffer [E, A, t, :]   (the buffer of learner.DeviceBatch, written in place)
    1 act kernel        reads the step's rows of the buffer in place (row stride T), writes action / log-prob in place
    1 async D2H copy    actions -> the workers' shared action block, 1 stream synchronise, 1 token per worker

-- no per-step allocation, no pageable copy, no collation: when the last env finishes the batch is already on the device in the
learner's layout; padded steps of shorter episodes are zeroed by four masked device ops (the reference zero-pads, :113-132).
"""
import numpy as np
import torch

from . import _native as N
from .learner import DeviceBatch


def _pin(arr):
    """Page-lock a numpy array's memory in place (hipHostRegister).  Returns True when the block is pinned."""
    rt = torch.cuda.cudart()
    rc = rt.cudaHostRegister(arr.ctypes.data, arr.nbytes, 0)
    return int(rc) == 0


def _unpin(arr):
    try:
        torch.cuda.cudart().cudaHostUnregister(arr.ctypes.data)
    except Exception:  # interpreter shutdown
        pass


class PinnedHostRollout:
    """One episode per env from a `ShmVectorEnv`, written straight into a device-resident DeviceBatch."""

    def __init__(self, venv, learner, recurrent, device, row_offset=0, t_cap=64, pad=None):
        self.v, self.L, self.recurrent, self.dev = venv, learner, recurrent, device
        self.lib = N.load()
        self.E, self.A, self.Do, self.Ds, self.K = venv.E, venv.A, venv.Do, venv.Ds, venv.K
        self.row_offset = int(row_offset)
        a = venv.arr
        self.pinned = all([_pin(a[k]) for k in ("obs", "state", "avail", "actions")])
        if not self.pinned:
            raise N.NativeError("hipHostRegister of the shared env blocks failed: use --vector_env=shm (host-staged copies)")
        self.h_obs, self.h_state = torch.from_numpy(a["obs"]), torch.from_numpy(a["state"])
        self.h_avail, self.h_act = torch.from_numpy(a["avail"]), torch.from_numpy(a["actions"])
        self.cap = int(t_cap)
        # leading dimensions rounded up to 4 floats for the MLP learners' 16-byte tile loads; the GRU / COMA kernels read contiguous rows
        self.pad = (not recurrent) if pad is None else bool(pad)
        self.buf = DeviceBatch(self.E, self.A, self.cap, self.Do, self.Ds, self.K, device, pad_obs=self.pad, pad_state=self.pad)
        self.h = None
        self.ws = None
        self.calls = 0

    def close(self):
        if self.pinned:
            for k in ("obs", "state", "avail", "actions"):
                _unpin(self.v.arr[k])
            self.pinned = False

    def _grow(self):
        old, T0 = self.buf, self.cap
        self.cap *= 2
        b = DeviceBatch(self.E, self.A, self.cap, self.Do, self.Ds, self.K, self.dev, pad_obs=self.pad, pad_state=self.pad)
        b.obs[:, :, :T0] = old.obs; b.state[:, :T0] = old.state; b.avail[:, :, :T0] = old.avail
        b.action[:, :, :T0] = old.action; b.logp[:, :, :T0] = old.logp
        self.buf = b

    def _act(self, t, seed, eps):
        """Sample the actions of step t for ALL rows from the buffer in place (rows of finished envs are ignored by the workers)."""
        b, spec, lib, s = self.buf, self.L.actor_spec, self.lib, N.stream_ptr()
        E, A, T, Do, K = self.E, self.A, self.cap, self.Do, self.K
        off = lambda ten, nbytes: N.C.c_void_p(ten.data_ptr() + nbytes)
        self.calls += 1
        if self.recurrent:
            N.check(lib.cm_gru_policy_act(off(b.obs, 4 * t * b.obs_ld), T * b.obs_ld, off(b.avail, t * K), T * K, E * A, spec.din, spec.hidden, K,
                                          N.ptr(self.L.actor), N.ptr(self.h), seed, self.row_offset, self.calls, off(b.action, 4 * t),
                                          off(b.logp, 4 * t), T, s), "cm_gru_policy_act")
        else:
            need = lib.cm_policy_act_workspace_bytes(E * A, spec.din, spec.hidden, spec.n_layers, K)  # 0 unless layered
            if need and (self.ws is None or self.ws.numel() < need):
                self.ws = torch.empty(need, dtype=torch.uint8, device=self.dev)
            N.check(lib.cm_policy_act_ws(off(b.obs, 4 * t * b.obs_ld), T * b.obs_ld, off(b.avail, t * K), T * K, E * A, spec.din, spec.hidden,
                                         spec.n_layers, K, N.ptr(self.L.actor), float(eps), seed, self.row_offset, self.calls,
                                         off(b.action, 4 * t), off(b.logp, 4 * t), T, N.ptr(self.ws) if need else None, need, s),
                    "cm_policy_act_ws")

    def collect(self, seed, eps=0.0):
        """Returns (DeviceBatch with T = longest episode, stats) -- what driver.host_rollout_shm returns, built on the device."""
        v, E, A = self.v, self.E, self.A
        a = v.arr
        v._all("reset")
        if self.recurrent:
            if self.h is None:
                self.h = torch.zeros(E * A, self.L.actor_spec.hidden, dtype=torch.float32, device=self.dev)
            self.h.zero_()  # h = None at the start of every episode (cleanmarl/mappo_lstm_multienvs.py:402)
        rew = np.zeros((E, self.cap), np.float32)
        ep_len = np.zeros(E, np.int64)
        ep_info = [None] * E
        alive = np.arange(E)
        stream = torch.cuda.current_stream()
        t = 0
        while alive.size:
            if t == self.cap:
                self._grow()
                rew = np.concatenate([rew, np.zeros_like(rew)], axis=1)
            b = self.buf
            b.obs[:, :, t].copy_(self.h_obs, non_blocking=True)
            b.state[:, t].copy_(self.h_state, non_blocking=True)
            b.avail[:, :, t].copy_(self.h_avail, non_blocking=True)
            self._act(t, seed, eps)
            self.h_act.copy_(b.action[:, :, t], non_blocking=True)
            stream.synchronize()  # the workers need the actions; this also fences the H2D copies before the blocks are rewritten
            for infos in v._all("step"):
                if infos:
                    for e, info in infos.items():
                        ep_info[e] = info
            rew[alive, t] = a["reward"][alive]
            ep_len[alive] += 1
            alive = alive[a["alive"][alive] == 1]
            t += 1
        T = int(ep_len.max())
        out = DeviceBatch(E, A, T, self.Do, self.Ds, self.K, self.dev, pad_obs=self.pad, pad_state=self.pad)
        b = self.buf
        ep = torch.from_numpy(ep_len).to(self.dev)
        m = (torch.arange(T, device=self.dev)[None, :] < ep[:, None])          # [E, T] valid steps
        ma = m[:, None, :]
        out.obs.copy_(b.obs[:, :, :T] * ma[..., None]); out.state.copy_(b.state[:, :T] * m[..., None])
        out.avail.copy_(b.avail[:, :, :T] * ma[..., None].to(torch.uint8))
        out.action.copy_(b.action[:, :, :T] * ma.to(torch.int32)); out.logp.copy_(b.logp[:, :, :T] * ma)
        mask_np = np.arange(T)[None, :] < ep_len[:, None]
        out.reward.copy_(torch.from_numpy(rew[:, :T] * mask_np))
        out.ep_len.copy_(ep.to(torch.int32))
        stats = dict(ep_reward=(rew[:, :T] * mask_np).sum(1).tolist(), ep_len=ep_len.tolist(), infos=ep_info)
        return out, stats
