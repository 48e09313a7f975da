"""The N > 1 path of bench.py is "replicas only": per-rank independent work, a start barrier and a MAX reduction of
the wall time. Covered here with world_size = 2 on CPU (gloo), rendezvous on 127.0.0.1."""
import os
import socket
import time

import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


class _CpuStandInBackend:
    """The three members bench.rank_pass() needs from the product, served by the CPU oracle at a small N: a filter with run_prepared(),
    the prepared-frames container, a sync. Everything else (per-rank world and seed, frame flattening, warm-up, barriers, timed
    region, MAX over ranks) is bench.py's own code."""

    class _Prepared:
        def __init__(self, cam, imu_counts, imu_all, stamps, meas_counts, ids_all, y_all):
            self.cam, self.stamps = cam, stamps
            self.imu = np.split(imu_all.reshape(-1, 13), np.cumsum(imu_counts)[:-1])
            cuts = np.cumsum(meas_counts)[:-1]
            self.ids, self.y = np.split(ids_all, cuts), np.split(y_all.reshape(-1, 2), cuts)

    class _Filter:
        def __init__(self, orc):
            self.orc, self.frames_done = orc, 0

        def run_prepared(self, pf, start, count):
            for f in range(start, start + count):
                for s in pf.imu[f]:
                    self.orc.process_imu(s)
                self.orc.process_vision(pf.stamps[f], pf.cam, pf.ids[f], pf.y[f].reshape(-1))
                self.frames_done += 1
            return count

    def make_filter(self, settings, N, sensor, ids, p, t):
        from oracle_binding import OracleFilter

        return self._Filter(OracleFilter(settings, sensor, ids, p, t))

    def prepare(self, cam, *flat):
        return self._Prepared(cam, *flat)

    def sync(self, flt):
        pass

    def spin_up(self, flt):
        pass


def _bench_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import argparse

    import bench

    dist_ = bench.init_control_group(world)
    assert dist_.get_backend() == "gloo"  # the control group never creates an RCCL communicator
    args = argparse.Namespace(landmarks=8, warmup=2, steps=6)
    value, slowest, flt, wld, frames, _, _ = bench.rank_pass(args, rank, world, dist_, _CpuStandInBackend())
    est = flt.orc.state_estimate()[0]
    q.put((rank, value, slowest, flt.frames_done, wld.seed if hasattr(wld, "seed") else None, est[10:13].tolist()))
    dist_.destroy_process_group()


def test_bench_rank_path_under_gloo_with_a_stand_in_filter():
    """bench.py's own per-rank path (rank_pass: world seeded by the rank, filter, prepared frames, warm-up, barrier-bracketed timed
    region, MAX over ranks on the gloo control group) with world_size 2 on CPU. The filter is a CPU stand-in (the oracle at N = 8)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, v0, s0, n0, _, pos0), (r1, v1, s1, n1, _, pos1) = res
    assert (r0, r1) == (0, 1)
    assert n0 == n1 == 8  # warm-up + timed frames, every rank its own
    assert v0 == pytest.approx(v1) and s0 == pytest.approx(s1)  # same aggregate everywhere
    assert v0 == pytest.approx(2 * 6 / s0)  # units of ALL ranks over the slowest rank's time
    assert pos0 != pos1  # different seeds: the replicas really are independent filters on different data


def test_single_process_needs_no_process_group():
    from eqvio_amd.replicas import timed_replica_run

    value, slowest, mine = timed_replica_run(lambda: time.sleep(0.01) or 5, lambda: None, 5)
    assert slowest == mine and value == pytest.approx(5 / slowest)


class BenchStandIn(_CpuStandInBackend):
    """EQVIO_BENCH_BACKEND=test_replicas_gloo:BenchStandIn - what bench.py's main() loads instead of the HIP backend in the test below."""


def test_bench_py_launches_its_own_ranks():
    """`python bench.py --gpus 2`, plain, no torchrun (the way the driver started the 1-GPU bench last round): bench.py must start the two ranks
    itself and report n_gpus = 2. CPU stand-in filter; everything else is the real main()."""
    import json
    import subprocess

    env = dict(os.environ, EQVIO_BENCH_BACKEND="test_replicas_gloo:BenchStandIn", PYTHONPATH=os.path.join(ROOT, "tests") + os.pathsep + ROOT)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--landmarks", "8"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout  # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["launcher"] == "self" and out["steps"] == 5
    assert out["value"] == pytest.approx(2 * 5 / (out["ms_per_step"] * 5e-3))
    assert out["per_rank_updates_per_s"]["min"] <= out["value"] / 2 * 1.0001 and out["per_rank_updates_per_s"]["max"] >= out["per_rank_updates_per_s"]["min"]
    # a launcher that disagrees with --gpus is an error, not a silently smaller run
    bad = subprocess.run(cmd, env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "must agree" in bad.stderr
