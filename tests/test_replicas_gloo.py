"""The N > 1 path of bench.py is "replicas only": per-rank independent work, a start barrier and a MAX reduction of
the wall time. Covered here with world_size = 2 on CPU (gloo), rendezvous on 127.0.0.1."""
import os
import socket
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from eqvio_amd.replicas import timed_replica_run

    steps = 20
    per_step = 0.002 * (1 + rank)  # rank 1 is twice as slow: the aggregate must be priced on the slowest rank

    def run():
        for _ in range(steps):
            time.sleep(per_step)
        return steps

    value, slowest, mine = timed_replica_run(run, lambda: None, steps, dist=dist)
    q.put((rank, value, slowest, mine))
    dist.destroy_process_group()


def test_two_replicas_report_aggregate_over_slowest_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, v0, s0, m0), (r1, v1, s1, m1) = res
    assert v0 == pytest.approx(v1) and s0 == pytest.approx(s1)  # every rank sees the same aggregate
    assert s0 == pytest.approx(max(m0, m1))
    assert m1 > m0
    assert v0 == pytest.approx(2 * 20 / s0)
    assert 0.04 <= s0 < 0.5


def test_single_process_needs_no_process_group():
    from eqvio_amd.replicas import timed_replica_run

    value, slowest, mine = timed_replica_run(lambda: time.sleep(0.01) or 5, lambda: None, 5)
    assert slowest == mine and value == pytest.approx(5 / slowest)
