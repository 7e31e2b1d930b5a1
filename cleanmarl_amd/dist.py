"""Env-sharded data parallelism: one process per GPU, replicated parameters, environments
[rank*E/G, (rank+1)*E/G) per rank.  Because the reference's update is ONE full-batch step per epoch with all
loss terms being masked SUMS divided by N = b_mask.sum() (cleanmarl/mappo_multienvs.py:572-576), summing the
per-shard un-normalised gradient/statistic buffers and dividing by the global N reproduces the single-device
step exactly (up to fp32 re-association).  These helpers are device-agnostic (RCCL on GPUs, gloo in CPU tests).
"""
import os

import torch


def shard(n_envs, rank, world):
    """(first_env, n_local) of this rank's contiguous env shard."""
    base, rem = divmod(n_envs, world)
    return rank * base + min(rank, rem), base + (1 if rank < rem else 0)


def allreduce_sum_(buf, pg=None, world=1):
    """THE data-path collective: all-reduce(sum) of a [grads | stats] buffer per optimiser step (latency-bound 30 - 150 KB
    messages; RCCL picks a one-shot/tree algorithm over xGMI).  Blocking form: GRU chunk steps, COMA, single passes."""
    if world > 1:
        torch.distributed.all_reduce(buf, group=pg)
    return buf


def allreduce_sum_async(buf, pg=None):
    """Same collective, not waited for: returns the work handle (`.wait()` makes the CURRENT stream wait, not the host, on RCCL).
    PPOLearner uses it to put the actor's message under the critic pass and the critic's under the next actor pass."""
    return torch.distributed.all_reduce(buf, group=pg, async_op=True)


def merge_moments_(mom, pg=None, world=1):
    """mom = [count, mean, M2] (float64) of the local shard -> global triple (Chan et al. parallel variance),
    so that the unbiased std of cleanmarl/mappo_multienvs.py:143-146, 505-512 is computed over ALL envs."""
    if world > 1:
        parts = [torch.zeros_like(mom) for _ in range(world)]
        torch.distributed.all_gather(parts, mom, group=pg)
        n = sum(p[0] for p in parts)
        mean = sum(p[0] * p[1] for p in parts) / n
        m2 = sum(p[2] + p[0] * (p[1] - mean) ** 2 for p in parts)
        mom.copy_(torch.stack([n, mean, m2]))
    return mom


class PeerAllReduce:
    """One-shot peer all-reduce + optimiser step over hipIpc-mapped mailboxes (csrc/cm_peer.hip, include/cleanmarl_hip.h): the
    data-path replacement of `all_reduce(buf); optimiser step` for ONE gradient buffer of `n_floats` floats.  torch.distributed is used
    once, at construction, to exchange the mailbox handles (control plane); afterwards a step is two launches on the caller's stream
    and no library call.  One instance per network (the critic's steps run on their own stream)."""

    def __init__(self, n_floats, process_group=None):
        import ctypes as C
        from . import _native as N
        self.lib = lib = N.load()
        self.pg = process_group
        self.rank = torch.distributed.get_rank(process_group)
        self.world = torch.distributed.get_world_size(process_group)
        if self.world > 16:
            raise N.NativeError("PeerAllReduce: at most 16 ranks")
        self.n = int(n_floats)
        nbytes = lib.cm_peer_mailbox_bytes(self.world, self.n)
        own, handle = C.c_void_p(), C.create_string_buffer(lib.cm_peer_handle_bytes())
        N.check(lib.cm_peer_mailbox_alloc(nbytes, C.byref(own), handle), "cm_peer_mailbox_alloc")
        self.own = own
        handles = [None] * self.world
        torch.distributed.all_gather_object(handles, (os.getpid(), handle.raw), group=process_group)
        self.boxes = (C.c_void_p * self.world)()
        self._opened = []
        for r, (pid, raw) in enumerate(handles):
            if r == self.rank:
                self.boxes[r] = own
            else:
                p = C.c_void_p()
                N.check(lib.cm_peer_mailbox_open(C.create_string_buffer(raw, len(raw)), C.byref(p)), f"cm_peer_mailbox_open(rank {r})")
                self.boxes[r] = p
                self._opened.append(p)
        self.seq = 0
        # wall-time bound of the step kernel's wait for the peers' tags (CM_PEER_TIMEOUT_S, default 30 s: a slow peer -- checkpoint, host-env
        # stall, module load -- is healthy) and the page-locked word a step whose wait ran out reports through: the kernel SKIPS that step and
        # writes its seq here; check() raises on it (no host sync: the word lives in host memory)
        self.timeout_s = float(os.environ.get("CM_PEER_TIMEOUT_S", "30"))
        self.status = torch.zeros(1, dtype=torch.int32).pin_memory()
        torch.distributed.barrier(group=process_group)  # every mailbox is mapped everywhere before the first push

    def check(self):
        """Raise if a step of this mailbox gave up waiting for its peers (the kernel skipped it: this rank's parameters are one step behind
        the others' -- the run cannot continue; restart from the last checkpoint)."""
        bad = int(self.status[0])
        if bad:
            from . import _native as N
            raise N.NativeError(f"PeerAllReduce: rank {self.rank} waited more than {self.timeout_s:g} s for the gradient slots of optimiser step "
                                f"{bad} (a peer died, or its mailbox mapping does not reach this GPU); that step was skipped, the replicas "
                                "have diverged -- restart from the last checkpoint (CM_PEER_TIMEOUT_S raises the bound)")

    def step(self, buf, n_params, opt_step, stream_ptr):
        """push `buf` ([n_params + 8] floats) to every mailbox, then the fused fold + optimiser step on the own mailbox."""
        from . import _native as N
        self.check()  # a step that gave up earlier: stop before anything else is pushed
        self.seq += 1
        N.check(self.lib.cm_peer_push(N.ptr(buf), self.n, self.rank, self.world, self.boxes, self.seq, stream_ptr), "cm_peer_push")
        N.check(self.lib.cm_optimizer_step_peer(N.ptr(buf), n_params, self.own, self.world, self.seq, opt_step, self.timeout_s,
                                                N.ptr(self.status), stream_ptr), "cm_optimizer_step_peer")

    def close(self):
        """Unmap the peers' mailboxes and free the own one -- after a barrier: nobody may still push into a freed mailbox."""
        torch.cuda.synchronize()
        torch.distributed.barrier(group=self.pg)
        for p in self._opened:
            self.lib.cm_peer_mailbox_close(p)
        self._opened = []
        torch.distributed.barrier(group=self.pg)
        if self.own is not None:
            self.lib.cm_peer_mailbox_free(self.own)
            self.own = None
