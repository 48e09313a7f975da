"""Multi-GPU mode of the EqF path: REPLICAS ONLY. One independent filter per GPU / per process, no
collective on the data path; torch.distributed is used for the start barrier and for the max-over-ranks wall time that
bench.py reports. Works with backend "nccl" (RCCL, one rank per GPU) and "gloo" (CPU tests)."""
import time


def timed_replica_run(run_steps, sync, steps, dist=None, device=None):
    """Barrier, run `run_steps()` (K steps of this rank's own filter), `sync()`, barrier; return
    (aggregate steps/s over all ranks, max-over-ranks seconds, this rank's seconds)."""
    import torch

    world = dist.get_world_size() if dist is not None else 1

    def barrier():
        sync()
        if dist is not None:
            dist.barrier()
        sync()

    barrier()
    t0 = time.perf_counter()
    done = run_steps()
    sync()
    mine = time.perf_counter() - t0
    if done is not None and done != steps:
        raise RuntimeError(f"replica processed {done} of {steps} steps")
    slowest = mine
    if dist is not None:
        t = torch.tensor([mine], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        slowest = float(t.item())
    barrier()
    return steps * world / slowest, slowest, mine
