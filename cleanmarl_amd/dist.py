"""Env-sharded data parallelism: one process per GPU, replicated parameters, environments
[rank*E/G, (rank+1)*E/G) per rank.  Because the reference's update is ONE full-batch step per epoch with all
loss terms being masked SUMS divided by N = b_mask.sum() (cleanmarl/mappo_multienvs.py:572-576), summing the
per-shard un-normalised gradient/statistic buffers and dividing by the global N reproduces the single-device
step exactly (up to fp32 re-association).  These helpers are device-agnostic (RCCL on GPUs, gloo in CPU tests).
"""
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
