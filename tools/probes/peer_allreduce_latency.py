"""Stream-side cost of one exchange + optimiser step: one-shot peer all-reduce (csrc/cm_peer.hip: push + fused fold / step, two launches)
vs RCCL all-reduce + the fused step launch, for the actor's 33 KB and the critic's 116 KB message of config 3.  Two processes share
cuda:0 for the peer path (hipIpc between processes on one device; the xGMI hop of a real multi-GPU node is NOT in this number); RCCL is
timed with one rank (it refuses two ranks on one device), i.e. its launch / protocol floor."""
import os
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def worker(rank, world, port, backend):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    if backend == "nccl":
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    from cleanmarl_amd import _native as N
    from cleanmarl_amd import dist
    from cleanmarl_amd.learner import _Adam
    lib = N.load()
    for name, n in (("actor 8141 params", 8141), ("critic 28865 params", 28865)):
        p = torch.randn(n, device=dev)
        opt = _Adam(n, 8e-4, "Adam", dev)
        norm = torch.zeros(1, device=dev)
        buf = torch.randn(n + N.NUM_STATS, device=dev)
        buf[n + N.STAT_COUNT] = 100.0
        s = N.stream_ptr()
        peer = dist.PeerAllReduce(n + N.NUM_STATS, torch.distributed.group.WORLD) if backend != "nccl" else None

        def one():
            o = opt.next_step(p, norm, -1.0)
            if peer is not None:
                peer.step(buf, n, o, s)
            else:
                torch.distributed.all_reduce(buf)
                N.check(lib.cm_optimizer_step(N.ptr(buf), n, o, s), "step")
        for _ in range(50):
            one()
        torch.cuda.synchronize(); torch.distributed.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300):
            one()
        e1.record(); torch.cuda.synchronize()
        if rank == 0:
            print(f"{'peer mailboxes, 2 processes on one GPU' if peer else 'RCCL, 1 rank'}: {name}: {1e3 * e0.elapsed_time(e1) / 300:.1f} us per exchange + step", flush=True)
        if peer is not None:
            peer.close()
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    import socket
    def port():
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0)); return s.getsockname()[1]
    mp.spawn(worker, args=(2, port(), "gloo"), nprocs=2, join=True)
    mp.spawn(worker, args=(1, port(), "nccl"), nprocs=1, join=True)
