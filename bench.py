#!/usr/bin/env python
"""bench.py — EqF vision updates/sec at N = 200 landmarks on MI355X (BASELINE.json metric).

A "step" is ONE VIOFilter::processVisionData (src/VIOFilter.cpp:194-241) at constant N: fast-Riccati propagate
(VIO_eqf.cpp:62-72) + k = 10 observer steps (:47-60) + outlier statistics + vision update (:105-135) +
invalid-landmark check, on one independent filter per GPU (replicas; the single-filter update does not shard, no
collective on the data path). Inputs (IMU samples + id'd feature tracks of a synthetic world, EuRoC-like pinhole
camera, InvDepth chart, fast Riccati — configs/EQVIO_config_EuRoC_stationary.yaml's eqf block) are generated
before the timed region; Sigma and the landmark arrays stay resident in HBM, only the per-frame measurement
(<= 5 KB) crosses PCIe inside the timed region, as it does for any caller of the filter API.

Prints ONE JSON line (rank 0). `value` = frames processed by all ranks / max-over-ranks wall time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

N_LANDMARKS = 200
FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X datasheet fp64 matrix peak; the measured issue-rate ceiling is reported too


def flops_propagate(n):  # BASELINE.md §2 / SURVEY.md §8(d): dense formulation
    return 4.0 * n**3 + 24.0 * n**2 + 288.0 * n


def flops_update(n, m):
    return 4.0 * n * n * m + 4.0 * n * m * m + m**3 / 3.0 + 2.0 * n * m


def eurocish_settings():
    from eqvio_amd.capi import COORD_INVDEPTH, Settings

    s = Settings.defaults()
    # The structure of the eqf block of configs/EQVIO_config_EuRoC_stationary.yaml:17-56 (InvDepth chart, fast Riccati, discrete velocity
    # lift, continuous innovation lift, equivariant output, fixed scene depth 5.0, pixel noise 1.93, process / IMU noise values rounded),
    # with these values REPLACED for the synthetic steady-state workload: initialPointVariance 1.0 (shipped 129.9),
    # initialBiasOmegaVariance 0.01 (shipped 97162.8), outlierThresholdAbs / Prob 1e8 / 1e8 (shipped 4.852 / 0.0323): with the thresholds
    # off no landmark ever becomes an outlier candidate and, in the 'hover' world, none enters or leaves, so EVERY frame takes the
    # one-round-trip path (speculative tail never cancelled). That is the best case; `frame_mix` in the output line reports the same
    # filter with landmark turnover and with the shipped thresholds.
    s.coordinateChoice = COORD_INVDEPTH
    s.fastRiccati = 1
    s.useDiscreteInnovationLift = 0
    s.useDiscreteVelocityLift = 1
    s.useEquivariantOutput = 1
    s.useMedianDepth = 0
    s.initialSceneDepth = 5.0
    s.initialAttitudeVariance, s.initialPositionVariance, s.initialVelocityVariance = 0.1357, 0.1, 8.97e-8
    s.initialBiasAccelVariance, s.initialBiasOmegaVariance = 1.58, 0.01
    s.initialCameraAttitudeVariance, s.initialCameraPositionVariance = 0.00102, 0.0235
    s.initialPointVariance = 1.0
    s.measurementNoise, s.outlierThresholdAbs, s.outlierThresholdProb, s.featureRetention = 1.93, 1e8, 1e8, 0.186
    s.attitudeProcessVariance, s.positionProcessVariance, s.velocityProcessVariance = 6.03e-5, 9.98e-6, 0.0253
    s.biasAccelProcessVariance, s.biasOmegaProcessVariance = 0.0, 0.0
    s.cameraAttitudeProcessVariance, s.cameraPositionProcessVariance, s.pointProcessVariance = 5.08e-6, 1.22e-5, 2.98e-4
    s.velAccNoise, s.velAccBiasWalk, s.velGyrNoise, s.velGyrBiasWalk = 0.01244, 0.00446, 2.43e-4, 1.34e-4
    return s


def build_workload(seed, n_frames, N):
    from eqvio_amd.simworld import SimWorld

    world = SimWorld(seed=seed, num_points=N, max_features=N, trajectory="hover", noise_px=0.5)
    frames = list(world.frames(n_frames))
    return world, frames


def flatten_frames(frames):
    imu_counts = np.array([len(f[0]) for f in frames], np.int32)
    imu_all = np.concatenate([f[0] for f in frames]).reshape(-1)
    stamps = np.array([f[1] for f in frames])
    meas_counts = np.array([len(f[2]) for f in frames], np.int32)
    ids_all = np.concatenate([f[2] for f in frames]).astype(np.int32)
    y_all = np.concatenate([f[3] for f in frames])
    return imu_counts, imu_all, stamps, meas_counts, ids_all, y_all


def make_filter(world, settings, N, device, frames, Filter):
    # initial condition: the true state at t = 0 restricted to the tracked ids, landmarks perturbed (VIOSimulator::getFullState
    # with initialNoise, src/VIOSimulator.cpp:300-307, chart-space noise replaced by a plain Euclidean perturbation)
    ids0 = frames[0][2]
    sensor, ids, p = world.true_state(0.0, ids0)
    rng = np.random.default_rng(1234)
    p = p * (1.0 + 0.05 * rng.normal(size=(len(ids), 1)))
    return Filter(settings, sensor, ids, p, 0.0)


def _oracle_warm(world, frames, settings):
    from oracle_binding import OracleFilter

    ids0 = frames[0][2]
    sensor, ids, p = world.true_state(0.0, ids0)
    orc = OracleFilter(settings, sensor, ids, p, 0.0)
    imus, stamp, mid, y = frames[0]
    # warm the state with one real frame so Sigma is dense
    for s in range(len(imus)):
        orc.process_imu(imus[s])
    orc.process_vision(stamp, world.cam, mid, y)
    return orc


def _cpu_batch_worker(idx, core, N, reps):
    """One CPU oracle filter pinned to one core (the reference filter is single-threaded: one sequence per core is how a CPU host
    would run the 8-sequence batch of BASELINE.json configs[3]). Child process of cpu_batch_baseline: prints READY, waits for a line on
    stdin, times `reps` frames, prints the seconds."""
    from oracle_binding import ARITH_EFFICIENT

    try:
        os.sched_setaffinity(0, {core})
    except OSError:
        pass
    world, frames = build_workload(seed=900 + idx, n_frames=3, N=N)
    orc = _oracle_warm(world, frames, eurocish_settings())
    imus, stamp, mid, y = frames[1]
    dts = np.full(len(imus), 1.0 / world.imu_freq)
    print("READY", flush=True)
    sys.stdin.readline()
    t0 = time.perf_counter()
    orc.bench_frame(imus, dts, stamp, world.cam, mid, y, ARITH_EFFICIENT, reps)
    print("T %.9f" % (time.perf_counter() - t0), flush=True)


def cpu_batch_baseline(N, reps=3, max_procs=None, timeout=300.0):
    """One-filter-per-core CPU batch: C independent oracle filters, one process pinned to each core this process may
    run on, released together; aggregate = C * reps frames / slowest process."""
    import subprocess

    try:
        cores = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cores = list(range(os.cpu_count() or 1))
    if max_procs:
        cores = cores[:max_procs]
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-batch-worker", str(i), str(c), str(N), str(reps)], stdin=subprocess.PIPE,
                              stdout=subprocess.PIPE, text=True) for i, c in enumerate(cores)]
    try:
        deadline = time.time() + timeout
        for pr in procs:
            line = pr.stdout.readline()
            if line.strip() != "READY" or time.time() > deadline:
                raise RuntimeError("CPU batch worker did not get ready: %r" % line)
        for pr in procs:
            pr.stdin.write("GO\n")
            pr.stdin.flush()
        times = []
        for pr in procs:
            line = pr.stdout.readline().split()
            if len(line) != 2 or line[0] != "T":
                raise RuntimeError("CPU batch worker died")
            times.append(float(line[1]))
    finally:
        for pr in procs:
            try:
                pr.stdin.close()
                pr.wait(timeout=30)
            except Exception:
                pr.kill()
    return {"value": len(cores) * reps / max(times), "unit": "updates/s aggregate", "cores": len(cores), "frames_each": reps,
            "slowest_s": max(times), "fastest_s": min(times),
            "note": "one single-threaded oracle filter ('efficient dense' arithmetic) per host core, released together; the CPU column for the one-filter-per-GPU line"}


def cpu_baseline(world, frames, settings, N, batch=True):
    """Oracle (CPU restatement of the reference arithmetic) timed on this host: one core, bounded sample; then one filter per core."""
    from oracle_binding import ARITH_AS_WRITTEN, ARITH_EFFICIENT

    orc = _oracle_warm(world, frames, settings)
    imus, stamp, mid, y = frames[1]
    dts = np.full(len(imus), 1.0 / world.imu_freq)
    t_eff = orc.bench_frame(imus, dts, stamp, world.cam, mid, y, ARITH_EFFICIENT, 3)
    t_asw = orc.bench_frame(imus, dts, stamp, world.cam, mid, y, ARITH_AS_WRITTEN, 1)
    out = {
        "value": 1.0 / t_eff,
        "unit": "updates/s",
        "cores": 1,
        "kind": "port",
        "sample": f"oracle (plain C++ -O3 -march=native restatement of VIO_eqf.cpp:62-135), N={N}: 3 frames 'efficient dense' arithmetic "
        f"({t_eff * 1e3:.1f} ms/frame), 1 frame 'as written' (LU inverse, K evaluated twice, (K C) Sigma: {t_asw * 1e3:.1f} ms/frame = {1.0 / t_asw:.2f} updates/s); "
        f"host has {os.cpu_count()} cores, the reference filter is single-threaded",
    }
    if batch:
        try:
            out["one_filter_per_core"] = cpu_batch_baseline(N)
        except Exception as e:  # informational leg: never lose the bench line over it
            out["one_filter_per_core"] = {"error": repr(e)}
    return out


class HipBackend:
    """What a rank needs from the product: a filter on this rank's GPU, the prepared input containers, a device sync. The world_size-2 gloo
    test (tests/test_replicas_gloo.py) passes a CPU stand-in with the same three members and runs everything else in rank_pass() as is."""

    def __init__(self, local_rank):
        import torch

        from eqvio_amd.capi import load_eqf_lib

        assert torch.cuda.is_available(), "bench.py needs an MI355X: the EqF path has no CPU fallback"
        torch.cuda.set_device(local_rank)
        self.torch, self.local_rank, self.lib = torch, local_rank, load_eqf_lib()

    def make_filter(self, settings, N, sensor, ids, p, t):
        from eqvio_amd.capi import VIOFilter

        return VIOFilter(settings, max_landmarks=N, device=self.local_rank, sensor=sensor, ids=ids, p=p, time=t)

    def prepare(self, cam, *flat):
        from eqvio_amd.capi import PreparedFrames

        return PreparedFrames(cam, *flat)

    def sync(self, flt):
        self.lib.eqf_synchronize(flt.core_handle())
        self.torch.cuda.synchronize()

    def spin_up(self, flt, seconds=float(os.environ.get("EQVIO_BENCH_SPIN_UP_S", "0.4"))):
        """The GPU clocks down while the host builds the synthetic world (seconds of numpy) and takes > 100 ms to come back (measured:
        tests/run_configs.py config 3 read 1200-1900 instead of 2280 updates/s behind a multi-second idle). A short run, e.g. --steps 20, would
        time the ramp, not the filter: keep the device busy with the library's own fp64 MFMA micro-benchmark for a moment before the W warm-up
        frames. Not filter work, not timed, and independent of W."""
        import ctypes as C

        t = C.c_double()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            self.lib.eqf_mfma_f64_peak(flt.core_handle(), C.byref(t))


def init_control_group(world_size):
    """Replicas only: the ranks share NOTHING on the data path. The process group exists for the start/stop barrier and the MAX over the
    ranks' wall times, both on CPU tensors over gloo, so no RCCL communicator is ever created (nothing to mis-price against xGMI)."""
    if world_size <= 1:
        return None
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # gloo announces its connections on STDOUT ("[Gloo] Rank 0 is connected to ..."): keep the process's stdout to the one JSON line
    sys.stdout.flush()
    saved, null = os.dup(1), os.open(os.devnull, os.O_WRONLY)
    os.dup2(null, 1)
    try:
        dist.init_process_group("gloo")
        dist.barrier()
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)
        os.close(null)
    return dist


def rank_pass(args, rank, world_size, dist, backend, settings=None):
    """One rank's whole job: its own synthetic world (seeded by the rank), its own filter, W warm-up frames, K timed frames between
    barriers. Returns (aggregate updates/s over all ranks, max-over-ranks seconds, filter, world, frames, settings, this rank's seconds)."""
    from eqvio_amd.replicas import timed_replica_run

    N = args.landmarks
    settings = settings if settings is not None else eurocish_settings()
    world, frames = build_workload(seed=100 + rank, n_frames=args.warmup + args.steps + 2, N=N)
    assert all(len(f[2]) == N for f in frames), "the hover world keeps every tracked feature in view"
    flt = make_filter(world, settings, N, None, frames, lambda s, sensor, ids, p, t: backend.make_filter(s, N, sensor, ids, p, t))
    # The input containers (IMU samples, one VisionMeasurement = std::map of pixel coordinates per frame: what the reference's
    # tracker / data server hands to the filter) are built once, before the timed region; a step is processIMUData x k +
    # processVisionData on them. The measurement itself still crosses the C-ABI from host memory every frame.
    prepared = backend.prepare(world.cam, *flatten_frames(frames[: args.warmup + args.steps]))
    backend.spin_up(flt)
    if args.warmup:
        flt.run_prepared(prepared, 0, args.warmup)
    value, elapsed, mine = timed_replica_run(lambda: flt.run_prepared(prepared, args.warmup, args.steps), lambda: backend.sync(flt), args.steps, dist=dist)
    return value, elapsed, flt, world, frames, settings, mine


BATCH8 = [dict(trajectory=tr, maxFeatures=mf, seed=k) for k, (mf, tr) in enumerate((mf, tr) for mf in (40, 200) for tr in ("wave", "square", "sine", "line"))]


def batch8_sequence(q, duration=60.0):
    """Sequence q of BASELINE.json configs[3] ("batch of 8 EuRoC MH/V sequences, one filter instance per GPU") as SURVEY.md section 8(d) spells it out for a box without the
    datasets: the C++ SimulationDataServer (the reference's simulator restated, eqvio_amd/host/VIOSimulator.cpp) on trajectory BATCH8[q], 60 s at 200 Hz IMU / 20 Hz camera,
    pixel noise on, the EuRoC configuration's structure with simulator-consistent values (eqvio_amd/configs.py). Returns (settings, camera, initial sensor state, frames)."""
    from eqvio_amd.capi import SimSettings, SimulationDataServer
    from eqvio_amd.configs import euroc_settings, sim_consistent

    spec = BATCH8[q]
    fs = sim_consistent(euroc_settings(), measurementNoise=1.0)
    sim = SimSettings.defaults(duration=duration, trajectory=spec["trajectory"], numPoints=8000, wallDistance=3.0, numWalls=6, randomSeed=spec["seed"], maxFeatures=spec["maxFeatures"],
                               outputNoise=1, inputNoise=0)
    srv = SimulationDataServer(sim, fs)
    fs.cameraOffset[:] = srv.camera_offset()
    s0, _, _ = srv.true_state(0.0, True)
    frames, imus = [], []
    while srv.next_measurement_type() != srv.NONE:
        if srv.next_measurement_type() == srv.IMU:
            imus.append(srv.get_imu())
            continue
        stamp, ids, y = srv.get_vision()
        frames.append((np.array(imus).reshape(-1, 13), stamp, ids, y))
        imus = []
    return fs, srv.cam, s0, frames


def batch8(args, rank, world_size, dist, backend, stand_in, one_device, placement=None):
    """`bench.py --gpus N --config batch8`: the 8 sequences are shared out over the N ranks (rank r takes q = r, r + N, ...), one filter per sequence (main_opt-like: it starts
    without landmarks and adds / drops them itself), the input containers built before the timed region. value = vision updates of the whole batch / the slowest rank's time:
    total work is fixed, so this line is "scaling": "strong". Parity of two of the sequences against the oracle: tests/test_gpu_batch8.py."""
    from eqvio_amd.capi import VIOFilter

    mine_q = list(range(rank, len(BATCH8), world_size))
    jobs = []
    for q in mine_q:
        fs, cam, s0, frames = batch8_sequence(q)
        flt = VIOFilter(fs, max_landmarks=2 * BATCH8[q]["maxFeatures"] + 64, device=backend.local_rank if not stand_in else 0, sensor=s0, ids=np.zeros(0, np.int32), p=np.zeros((0, 3)), time=0.0)
        jobs.append((q, flt, backend.prepare(cam, *flatten_frames(frames)), len(frames)))
    if not stand_in and jobs:
        backend.spin_up(jobs[0][1])
    for _, flt, _, _ in jobs:
        backend.sync(flt)
    if dist is not None:
        dist.barrier()
    per_seq = {}
    t0 = time.perf_counter()
    for q, flt, prepared, nfr in jobs:
        t1 = time.perf_counter()
        done = flt.run_prepared(prepared, 0, nfr)
        backend.sync(flt)
        assert done == nfr, (q, done, nfr)
        per_seq[q] = {"frames": nfr, "seconds": time.perf_counter() - t1, "final_landmarks": (flt.sigma_dim() - 21) // 3}
    mine = time.perf_counter() - t0
    gathered = [(mine, per_seq, placement)]
    if dist is not None:
        gathered = [None] * world_size
        dist.all_gather_object(gathered, (mine, per_seq, placement))
        dist.barrier()
    if rank == 0:
        slowest = max(g[0] for g in gathered)
        seqs = {q: v for g in gathered for q, v in g[1].items()}
        total = sum(v["frames"] for v in seqs.values())
        rates = [v["frames"] / v["seconds"] for v in seqs.values()]
        print(json.dumps({
            "metric": "EqF vision updates/sec, batch of 8 sequences", "value": total / slowest, "unit": "updates/s", "n_gpus": world_size, "steps": total, "warmup": 0,
            "ms_per_step": 1e3 * slowest / total, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "stand-in" if stand_in else "synthetic",
            "config": {"workload": "BASELINE.json configs[3] stand-in: 8 simulated sequences (wave / square / sine / line x maxFeatures 40 / 200, seeds 0-7, 60 s, 200 Hz IMU, 20 Hz camera), "
                                   "EuRoC-structured settings, one filter per sequence, sequences shared out over the ranks" + (" (every rank on GPU 0: EQVIO_BENCH_ONE_DEVICE)" if one_device else ""),
                       "parallelism": f"replicas x{world_size}"},
            "per_rank_seconds": {"min": min(g[0] for g in gathered), "max": slowest},
            "placement": [g[2] for g in gathered],
            "per_sequence_updates_per_s": {"min": min(rates), "max": max(rates)},
            "sequences": {str(q): dict(BATCH8[q], **{k: (round(v, 4) if isinstance(v, float) else v) for k, v in seqs[q].items()}) for q in sorted(seqs)},
        }))
    if dist is not None:
        dist.destroy_process_group()


def load_backend(local_rank):
    """The product backend, or - test hook, CPU rehearsal of the launch logic only - the class named by EQVIO_BENCH_BACKEND=module:Class
    (tests/test_replicas_gloo.py passes its oracle-backed stand-in; such a run prints `"data": "stand-in"` and no roofline)."""
    spec = os.environ.get("EQVIO_BENCH_BACKEND")
    if not spec:
        return HipBackend(local_rank), False
    import importlib

    mod, cls = spec.split(":")
    return getattr(importlib.import_module(mod), cls)(), True


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks here (what torch.distributed.run would do: RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in the environment, rendezvous on 127.0.0.1), pass rank 0's one JSON line through, fail if any rank fails."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), EQVIO_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out)
    sys.stdout.flush()
    if any(codes):
        sys.stderr.write("bench.py --gpus %d: rank exit codes %r\n" % (n, codes))
        sys.exit(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--landmarks", type=int, default=N_LANDMARKS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-multi-filter", action="store_true")
    ap.add_argument("--no-frame-mix", action="store_true")
    ap.add_argument("--no-binding", action="store_true")
    ap.add_argument("--no-sizes", action="store_true", help="skip the other state sizes (BASELINE configs 2 / 3: 40 and 500 landmarks)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 counter passes of the roofline object (three short runs of this script under the profiler)")
    ap.add_argument("--config", default="headline", choices=["headline", "batch8"],
                    help="batch8: BASELINE.json configs[3] - a batch of 8 sequences (trajectories wave / square / sine / line x maxFeatures 40 / 200, seeds 0-7, 60 s each) "
                         "shared out over the ranks, one filter per sequence; --steps / --warmup / --landmarks are ignored")
    args = ap.parse_args()

    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:  # plain `python bench.py --gpus N`: this process is the launcher
            return self_launch(args.gpus)
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.exit("bench.py: launched with WORLD_SIZE=%s but --gpus %d: the rank count and --gpus must agree" % (os.environ["WORLD_SIZE"], args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = args.gpus
    one_device = bool(os.environ.get("EQVIO_BENCH_ONE_DEVICE"))  # test hook: every rank on GPU 0 (rehearsal of the N > 1 launch on a 1-GPU box)
    # Every rank's host threads go onto physical cores of its GPU's NUMA node BEFORE the HIP context (and with it the pinned doorbell / result packets) exists:
    # the frame boundary runs through the host (eqvio_amd/placement.py). Restored before the CPU legs of the single-rank run, which want every core.
    all_cpus = os.sched_getaffinity(0)
    placement = None
    if not os.environ.get("EQVIO_BENCH_BACKEND") and not os.environ.get("EQVIO_BENCH_NO_PIN"):
        from eqvio_amd.placement import pin_rank

        placement = pin_rank(local_rank, world_size, devices=[0] * world_size if one_device else None)
    if one_device:
        local_rank = 0
    backend, stand_in = load_backend(local_rank)
    if not stand_in and not one_device:
        have = backend.torch.cuda.device_count()
        assert have >= world_size, "bench.py --gpus %d: only %d device(s) visible" % (world_size, have)
    dist = init_control_group(world_size)
    if args.config == "batch8":
        return batch8(args, rank, world_size, dist, backend, stand_in, one_device, placement)
    N = args.landmarks

    def Filter(settings, sensor, ids, p, t):
        return backend.make_filter(settings, N, sensor, ids, p, t)

    value, elapsed, flt, world, frames, settings, mine = rank_pass(args, rank, world_size, dist, backend)
    per_rank, placements = [mine], [placement]
    if dist is not None:  # every rank's own wall time of the timed region, for the min / max next to the aggregate
        per_rank, placements = [None] * world_size, [None] * world_size
        dist.all_gather_object(per_rank, mine)
        dist.all_gather_object(placements, placement)
    n, m = 21 + 3 * N, 2 * N
    ms_per_step = 1e3 * elapsed / args.steps
    frame_flops = flops_propagate(n) + flops_update(n, m)
    roofline = cpu = multi = mix = binding = factorisation = sizes = None
    if not stand_in:
        lib = backend.lib
        cam = world.cam
        core = flt.core_handle()
        # post-run sanity: the state is finite and Sigma is symmetric positive definite
        S = flt.get_sigma()
        assert np.all(np.isfinite(S)) and S.shape[0] == 21 + 3 * N
        assert np.linalg.eigvalsh(0.5 * (S + S.T)).min() > 0
        # which factorisation the timed frames really ran: look-ahead launches, and how many of them stalled and were redone on the launch chain
        import ctypes as C

        la_l, la_f = C.c_long(), C.c_long()
        assert lib.eqf_lookahead_stats(core, C.byref(la_l), C.byref(la_f), 0) == 0
        factorisation = {"lookahead_launches": la_l.value, "stalled_and_redone_on_the_chain": la_f.value, "frames": args.warmup + args.steps}
        if hasattr(lib, "eqf_lookahead_home"):  # EQF_OPT_LA_HOME: launches that kept the owner and the S half-rows on one XCD (a filter that has the device to itself)
            hx, hl = C.c_int(), C.c_long()
            assert lib.eqf_lookahead_home(core, C.byref(hx), C.byref(hl)) == 0
            factorisation.update({"home_placement_launches": hl.value, "home_xcd": hx.value})
        if hasattr(lib, "eqf_early_doorbell_stats"):  # EQF_OPT_EARLY_DOORBELL: updates the host took from the look-ahead kernel's own doorbell
            er = C.c_long()
            assert lib.eqf_early_doorbell_stats(core, C.byref(er), 0) == 0
            factorisation["updates_taken_from_the_early_doorbell"] = er.value
        if rank == 0 and not args.no_roofline:
            roofline = measure_roofline(flt, lib, core, cam, frames, args, n, m)
            if not args.no_pmc and world_size == 1 and not stand_in:
                roofline.update(live_pmc(N, roofline, args.steps / elapsed))
        if placement is not None and placement.get("pinned"):
            os.sched_setaffinity(0, all_cpus)  # the legs below start threads / processes of their own on every core
        if rank == 0 and world_size == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(world, frames, settings, N)
        if rank == 0 and world_size == 1 and not args.no_multi_filter:
            multi = several_filters_on_one_gpu(settings, N, local_rank, Filter, lib)
        if rank == 0 and world_size == 1 and not args.no_frame_mix:
            mix = frame_mix(N, local_rank, lib)
        if rank == 0 and world_size == 1 and not args.no_binding:
            binding = reference_side_binding(N)
        if rank == 0 and world_size == 1 and not args.no_sizes and not stand_in:
            sizes = other_state_sizes(local_rank, lib)

    if rank == 0:
        out = {
            "metric": "EqF vision updates/sec @ N=200 landmarks",
            "value": value,
            "unit": "updates/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "stand-in" if stand_in else "synthetic",
            "config": {
                "workload": f"synthetic 'hover' world, {N} tracked landmarks (state dim {n}, {m} measurement rows), InvDepth chart, fast Riccati, "
                "IMU 200 Hz / camera 20 Hz (10 observer steps per frame), EuRoC pinhole intrinsics, one independent filter per GPU (replicas)",
                "landmarks": N,
                "state_dim": n,
                "parallelism": f"replicas x{world_size}",
            },
            "per_rank_updates_per_s": {"min": args.steps / max(per_rank), "max": args.steps / min(per_rank)},
            "placement": placements,  # per rank: PCI bus id of its GPU, that GPU's NUMA node, the cores the rank was pinned to (eqvio_amd/placement.py)
            "launcher": "self" if os.environ.get("EQVIO_BENCH_SELF_LAUNCHED") else ("external" if "WORLD_SIZE" in os.environ else "single process"),
            "frame_dense_flops": frame_flops,
            "dense_equiv_tflops": frame_flops * value / world_size / 1e12,
            "dense_equiv_frac_of_fp64_mfma_peak": frame_flops * value / world_size / 1e12 / FP64_MFMA_PEAK_TFLOPS,
        }
        if factorisation is not None:
            out["factorisation"] = factorisation
        if roofline is not None:
            out["roofline"] = roofline
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if multi is not None:
            out["several_filters_one_gpu"] = multi
        if mix is not None:
            out["frame_mix"] = mix
        if binding is not None:
            out["reference_side_binding"] = binding
        if sizes is not None:
            out["other_state_sizes"] = sizes
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def other_state_sizes(device, lib, sizes=(40, 500), n_frames=300, n_warm=60):
    """The same hover workload at the sizes of BASELINE's other configurations - 40 landmarks (`maxFeatures` of the reference's shipped configurations, config 2's range) and 500
    (config 3) - through the same calls as the headline line: updates/s, the look-ahead kernel's launches (stalled ones are redone on the launch chain and counted) and how many of them
    built Z themselves (no k_build_Z launch: 3 launches per frame)."""
    import ctypes as C

    from eqvio_amd.capi import PreparedFrames, VIOFilter

    out = []
    for Nl in sizes:
        world, frames = build_workload(seed=500 + Nl, n_frames=n_warm + n_frames + 2, N=Nl)
        flt = make_filter(world, eurocish_settings(), Nl, device, frames, lambda s, se, i, p, t: VIOFilter(s, max_landmarks=Nl, device=device, sensor=se, ids=i, p=p, time=t))
        core = flt.core_handle()
        warm = PreparedFrames(world.cam, *flatten_frames(frames[:n_warm]))
        timed = PreparedFrames(world.cam, *flatten_frames(frames[n_warm:n_warm + n_frames]))
        assert flt.run_prepared(warm) == n_warm
        lib.eqf_synchronize(core)
        a, b, z = C.c_long(), C.c_long(), C.c_long()
        lib.eqf_lookahead_stats(core, C.byref(a), C.byref(b), 1)
        lib.eqf_z_in_lookahead_stats(core, C.byref(z), 1)
        t0 = time.perf_counter()
        assert flt.run_prepared(timed) == n_frames
        lib.eqf_synchronize(core)
        el = time.perf_counter() - t0
        lib.eqf_lookahead_stats(core, C.byref(a), C.byref(b), 0)
        lib.eqf_z_in_lookahead_stats(core, C.byref(z), 0)
        out.append({"landmarks": Nl, "state_dim": 21 + 3 * Nl, "value": n_frames / el, "unit": "updates/s", "us_per_frame": 1e6 * el / n_frames, "frames": n_frames,
                    "lookahead_launches": a.value, "stalled_and_redone_on_the_chain": b.value, "launches_that_built_Z_themselves": z.value})
        flt.close()
    return out


def reference_side_binding(N, n_frames=400):
    """The rate a maintainer's tree sees: the member sequence that the reference's VIOFilter::processVisionData makes on its VIO_eqf
    (src/VIOFilter.cpp:194-241), replayed over the reference-side binding tests/integration/VIO_eqf_mi355x.cpp by tests/integration/run_filter_frames.cpp (the
    state estimate read after every frame, as main_sim does), on the headline workload. (i) member for member: integrateRiccatiStateFast + k x
    integrateObserverState + one getOutputCovById per landmark + performVisionUpdate; (ii) fused: the two hunks of tests/integration/VIOFilter_mi355x_hunks.hpp
    (eqf_stage_measurement / eqf_propagate_fast / eqf_stats_then_update). Parity of both against the oracle, frame by frame: tests/test_integration_filter.py."""
    import tempfile

    from integration_scenario import build_driver, plan_without_decisions, run_driver, write_scenario

    try:
        build_driver()
        settings = eurocish_settings()
        world, frames = build_workload(seed=100, n_frames=n_frames + 1, N=N)
        ids0 = frames[0][2]
        sensor, ids, p = world.true_state(0.0, ids0)
        p = p * (1.0 + 0.05 * np.random.default_rng(1234).normal(size=(len(ids), 1)))
        out = {}
        with tempfile.TemporaryDirectory() as tmp:
            scen = os.path.join(tmp, "scenario.bin")
            write_scenario(scen, settings, world.cam, sensor, ids, p, 0.0, frames[:n_frames], plan_without_decisions(frames[:n_frames], 0.0))
            for key, fused, nfr in (("member_for_member", 0, min(n_frames, 120)), ("fused", 1, n_frames)):
                if nfr != n_frames:
                    write_scenario(scen + ".short", settings, world.cam, sensor, ids, p, 0.0, frames[:nfr], plan_without_decisions(frames[:nfr], 0.0))
                info = run_driver(scen if nfr == n_frames else scen + ".short", os.path.join(tmp, "out.bin"), fused, warm=20)
                out[key] = {"value": info["updates_per_s"], "unit": "updates/s", "frames": info["frames"]}
                if info.get("gain_matrix_seconds"):  # what the unmodified caller spends building its dense n x n / m x m gain matrices per frame (src/VIOFilter.cpp:155-158, :232): not the binding's
                    g = info["gain_matrix_seconds"]
                    out[key]["caller_side_gain_matrix_us_per_frame"] = 1e6 * g / info["frames"]
                    out[key]["value_without_the_callers_gain_matrices"] = info["frames"] / (info["seconds"] - g)
        out["note"] = ("reference-side VIO_eqf binding (tests/integration/VIO_eqf_mi355x.cpp): the member sequence of VIOFilter::processVisionData replayed + stateEstimate per frame, "
                       "N = %d hover world; the mirror's own rate is the headline value" % N)
        return out
    except Exception as e:  # informational leg: never lose the bench line over it
        return {"error": repr(e)}


def frame_mix(N, device, lib, n_frames=1200, n_warm=200):
    """The headline workload is the best case (see eurocish_settings). Here the same filter runs on the 'wave' world (SimWorld, the
    reference's wave trajectory, SimulationDataServer.cpp:46-65): ~9 of 200 tracked features change per frame (removeOldLandmarks +
    addNewLandmarks: Sigma compaction and growth, a second host round trip), first with the outlier thresholds off, then with the shipped
    EuRoC thresholds / retention / point variance (EQVIO_config_EuRoC_stationary.yaml:26-32: 4.852 px, 0.0323, 0.186, 129.9), where the
    statistics kernel cancels the speculative tail whenever a landmark is an outlier candidate (VIOFilter.cpp:304-364)."""
    import ctypes as C
    import time

    from eqvio_amd.capi import PreparedFrames, VIOFilter
    from eqvio_amd.simworld import SimWorld

    out = []
    for name, thr_abs, thr_prob, pvar in (("wave world, landmark turnover, outlier thresholds off", 1e8, 1e8, 1.0),
                                          ("wave world, landmark turnover, shipped EuRoC outlier thresholds 4.852 px / 0.0323, retention 0.186, point variance 129.9", 4.852186665580312,
                                           0.03229809583062128, 129.90415638150924)):
        s = eurocish_settings()
        s.outlierThresholdAbs, s.outlierThresholdProb, s.featureRetention, s.initialPointVariance = thr_abs, thr_prob, 0.18594708334486176, pvar
        world = SimWorld(seed=321, num_points=2500, max_features=N, trajectory="wave", noise_px=0.5)
        frames = list(world.frames(n_warm + n_frames))
        ids0 = frames[0][2]
        sensor, ids, p = world.true_state(0.0, ids0)
        p = p * (1.0 + 0.05 * np.random.default_rng(7).normal(size=(len(ids), 1)))
        flt = VIOFilter(s, max_landmarks=N + 120, device=device, sensor=sensor, ids=ids, p=p, time=0.0)
        core = flt.core_handle()
        prepared = PreparedFrames(world.cam, *flatten_frames(frames))
        turn = float(np.mean([len(set(a[2].tolist()) ^ set(b[2].tolist())) for a, b in zip(frames[n_warm:-1], frames[n_warm + 1:])]))
        flt.run_prepared(prepared, 0, n_warm)
        lib.eqf_synchronize(core)
        c0, q0, x0 = C.c_long(), C.c_long(), C.c_long()
        lib.eqf_speculation_stats(core, C.byref(c0), C.byref(q0), C.byref(x0), 1)
        dims = []
        t0 = time.perf_counter()
        for k in range(0, n_frames, 100):
            flt.run_prepared(prepared, n_warm + k, min(100, n_frames - k))
            dims.append(flt.sigma_dim())
        lib.eqf_synchronize(core)
        el = time.perf_counter() - t0
        lib.eqf_speculation_stats(core, C.byref(c0), C.byref(q0), C.byref(x0), 0)
        g0, h0 = C.c_long(), C.c_long()  # propagation launches (warm-up included) that applied a record of removed landmarks / created the frame's new landmarks themselves
        lib.eqf_gather_stats(core, C.byref(g0), 0)
        lib.eqf_hold_stats(core, C.byref(h0), 0)
        S = flt.get_sigma()
        ok = bool(np.all(np.isfinite(S)))
        flt.close()
        out.append({"workload": name, "value": n_frames / el, "unit": "updates/s", "frames": n_frames, "features_changed_per_frame": round(turn, 2),
                    "mean_landmarks": (float(np.mean(dims)) - 21.0) / 3.0, "frames_with_tail_queued_speculatively": q0.value / max(c0.value, 1),
                    "frames_with_tail_cancelled_on_device": x0.value / max(c0.value, 1),
                    "frames_whose_removed_landmarks_left_inside_the_propagation_kernel": g0.value / (n_warm + n_frames),
                    "frames_whose_new_landmarks_were_created_by_the_propagation_kernel": h0.value / (n_warm + n_frames), "sigma_finite": ok})
    return out


def several_filters_on_one_gpu(settings, N, device, Filter, lib, counts=(1, 2, 4, 8, 16), sizes=(50, 200), n_frames=300, n_warm=60):
    """"optionally several filters per GPU to fill CUs". A single filter's frame is a latency-bound dependent chain that leaves
    most CUs idle; R independent filters (one host thread, one eqf_ctx, one stream pair each) overlap on the same GPU. The sweep over R at N = 50
    and N = 200 locates the per-GPU saturation point ahead of the 8-GPU batch (BASELINE config 4 could run several sequences per GPU).
    Informational: the headline value stays one filter per GPU, as BASELINE.json's north_star says. Each persistent look-ahead kernel needs its
    workgroups co-resident; `lookahead_fallbacks` counts launches that stalled under the sharing and were redone on the launch chain."""
    import ctypes as C
    import threading
    import time

    from eqvio_amd.capi import PreparedFrames, VIOFilter

    def run_config(Nl, R):
        flts, work = [], []
        for r in range(R):
            world, frames = build_workload(seed=500 + r, n_frames=n_warm + n_frames + 2, N=Nl)
            flts.append(make_filter(world, settings, Nl, device, frames, lambda s, se, i, p, t: VIOFilter(s, max_landmarks=Nl, device=device, sensor=se, ids=i, p=p, time=t)))
            work.append(PreparedFrames(world.cam, *flatten_frames(frames[: n_warm + n_frames])))
        for f, pf in zip(flts, work):
            f.run_prepared(pf, 0, n_warm)
            lib.eqf_synchronize(f.core_handle())
            lib.eqf_lookahead_stats(f.core_handle(), None, None, 1)
        barrier = threading.Barrier(R + 1)
        errs = []

        def run(f, pf):
            barrier.wait()
            try:
                f.run_prepared(pf, n_warm, n_frames)  # ctypes releases the GIL: the host threads really run in parallel
                lib.eqf_synchronize(f.core_handle())
            except Exception as e:  # noqa: BLE001
                errs.append(repr(e))

        ths = [threading.Thread(target=run, args=(f, pf)) for f, pf in zip(flts, work)]
        for t in ths:
            t.start()
        barrier.wait()
        t0 = time.perf_counter()
        for t in ths:
            t.join()
        el = time.perf_counter() - t0
        la, fb = 0, 0
        for f in flts:
            a, b = C.c_long(), C.c_long()
            lib.eqf_lookahead_stats(f.core_handle(), C.byref(a), C.byref(b), 0)
            la, fb = la + a.value, fb + b.value
            f.close()
        out = {"filters": R, "value": R * n_frames / el, "lookahead_launches": la, "lookahead_fallbacks": fb}
        if errs:
            out["errors"] = errs
        return out

    # (N = 200: three and six filters as well - four look-ahead kernels of 65 workgroups do not fit 256 compute units, and R = 4 is a stable 20 k where R = 3 and R = 6 reach 28 - 29 k)
    sweep = {"N%d" % Nl: [run_config(Nl, R) for R in (sorted(set(counts) | {3, 6}) if Nl >= 200 else counts) if not (Nl >= 200 and R > 8)] for Nl in sizes}
    # N = 200 again with EQF_OWN_HW_QUEUES (eqf_hip.h: eqf_own_hardware_queue): a hardware queue per context instead of the runtime's four shared ones
    own = None
    try:
        os.environ["EQF_OWN_HW_QUEUES"] = "4"
        own = [run_config(N, R) for R in (2, 3, 4)]
    except Exception as e:  # noqa: BLE001
        own = {"error": repr(e)[:200]}
    finally:
        os.environ.pop("EQF_OWN_HW_QUEUES", None)
    # the same with one PROCESS per filter (the reference-compatible mode: its LoopTimer is a global, include/eqvio/LoopTimer.h:95): scripts/multi_process.py
    procs = None
    try:
        import subprocess

        res = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "multi_process.py"), str(N), "1500", "1,2,3,4,6"], capture_output=True, text=True, timeout=600, check=True)
        procs = {int(k): v for k, v in json.loads(res.stdout.strip().splitlines()[-1])["aggregate_updates_per_s"].items()}
    except Exception as e:  # noqa: BLE001
        procs = {"error": repr(e)[:200]}
    threads_best = max(sweep.get("N%d" % N, [{"value": 0.0, "filters": 0}]) + (own if isinstance(own, list) else []), key=lambda r: r["value"])
    best_p = max(((v, k) for k, v in procs.items() if isinstance(k, int)), default=(0.0, 0))
    return {"filters": best_p[1] if best_p[0] > threads_best["value"] else threads_best["filters"], "frames_each": n_frames, "value": max(best_p[0], threads_best["value"]),
            "unit": "updates/s aggregate on one GPU", "sweep": sweep, "threads_with_own_hardware_queues_N%d" % N: own, "one_process_per_filter_N%d" % N: procs,
            "note": "informational; sweep = aggregate updates/s for R independent filters as R threads of one process (one context + one stream each) at N = 50 / 200; "
                    "one_process_per_filter = the same as R processes. What bounds it (DESIGN.md section 7): (1) HARDWARE QUEUES - the runtime shares its GPU_MAX_HW_QUEUES = 4 "
                    "hardware queues among a process' plain streams, its own null stream included, and two filters whose streams share a queue take turns kernel by kernel: four "
                    "threads on the shared queues are a stable 20 k at N = 200 (exactly what GPU_MAX_HW_QUEUES = 2 gives), three reach 28 k or 16 k depending on which streams share "
                    "a queue; with a queue per context (EQF_OWN_HW_QUEUES=4: hipExtStreamCreateWithCUMask streams own their queue) R = 2 / 3 / 4 reach 20 / 28 / 32 k in one process "
                    "(threads_with_own_hardware_queues). (2) COMPUTE UNITS - a look-ahead kernel needs its 66 workgroups (N = 200) resident at once, one per compute unit, so three fit "
                    "a 256-CU device and a fourth takes turns (launches are booked against the CUs inside a process; across processes the bounded waits and the chain retry keep "
                    "every filter correct). Round 4's third explanation - the HIP runtime's launch path serialising the host threads - was WRONG (a launch baton that lets one "
                    "thread into the runtime at a time changed nothing: scripts/dbg/r05_variants/launch_baton.patch)"}


def measure_roofline(flt, lib, core, cam, frames, args, n, m):
    """Per-kernel HIP-event durations (events recorded on the context's stream around each launch) over a slice of the
    same workload, run right after the timed region. The dominant kernel family is priced against the fp64 MFMA peak
    with its ALGORITHMIC flops (dense formulation of SURVEY.md §8(d))."""
    import ctypes as C

    from eqvio_amd.capi import OPT_LIFT_WITH_SYRK, OPT_TIMING

    nfr = min(20, args.steps)
    sl = flatten_frames(frames[args.warmup + args.steps : args.warmup + args.steps + 2] * (nfr // 2))
    # re-stamp so that time keeps increasing
    t_last = flt.get_time()
    k = len(sl[2])
    sl = list(sl)
    sl[2] = t_last + (np.arange(k) + 1) * 0.05
    imu = sl[1].reshape(-1, 13)
    per = len(imu) // k
    for j in range(k):
        imu[j * per : (j + 1) * per, 0] = sl[2][j] - 0.05 + np.arange(per) * 0.005
    sl[1] = imu.reshape(-1)
    lws = C.c_int(0)
    lift_with_syrk = lib.eqf_get_option(core, OPT_LIFT_WITH_SYRK, C.byref(lws)) == 0 and lws.value != 0
    lib.eqf_set_option(core, OPT_TIMING, 1)
    flt.run_frames(cam, *sl)
    which = np.zeros(65536, np.int32)
    us = np.zeros(65536, np.float32)
    cnt = lib.eqf_last_kernel_times(core, which.ctypes.data_as(C.POINTER(C.c_int)), us.ctypes.data_as(C.POINTER(C.c_float)), len(us))
    lib.eqf_set_option(core, OPT_TIMING, 0)
    agg = {}
    for i in range(cnt):
        name = lib.eqf_kernel_name(int(which[i])).decode()
        agg.setdefault(name, []).append(float(us[i]))
    per_frame = {k_: sum(v) / k for k_, v in agg.items()}
    launches = {k_: len(v) / k for k_, v in agg.items()}
    # Round 6: a second pass over the same frames with ONLY the factorisation's launches bracketed (option 100 = 2): the frame around them is the timed region's own - lift and
    # covariance update one launch, early doorbell, no events between the other kernels - so the dominant kernel runs with the GPU as busy around it as it is there. (With every
    # kernel bracketed the pass came out 15 % slower in some process runs on some boxes, all kernels alike, while the timed region and rocprofv3 did not move:
    # profiles/r06_syrk_front_end_ab.txt.) `achieved` / `frac` are priced on this pass; the all-kernels pass stays in per_kernel_us_per_frame.
    sl2 = list(sl)
    sl2[2] = sl[2] + k * 0.05
    imu2 = sl[1].reshape(-1, 13).copy()
    imu2[:, 0] += k * 0.05
    sl2[1] = imu2.reshape(-1)
    dom_only = None
    if lib.eqf_set_option(core, OPT_TIMING, 2) == 0:
        flt.run_frames(cam, *sl2)
        cnt2 = lib.eqf_last_kernel_times(core, which.ctypes.data_as(C.POINTER(C.c_int)), us.ctypes.data_as(C.POINTER(C.c_float)), len(us))
        lib.eqf_set_option(core, OPT_TIMING, 0)
        agg2 = {}
        for i in range(cnt2):
            agg2.setdefault(lib.eqf_kernel_name(int(which[i])).decode(), []).append(float(us[i]))
        dom_only = {k_: sum(v) / k for k_, v in agg2.items()}
    tpeak = C.c_double()
    lib.eqf_mfma_f64_peak(core, C.byref(tpeak))
    # kernel families and their algorithmic (dense-formulation) flops per frame
    fam = {
        "cholesky+trsm factorisation of [S; T; y^T] (k_chol_lookahead: one persistent kernel per frame; or k_chol_step, one launch per panel; first tile inside k_build_Z)": (
            ["k_chol_first", "k_chol_step", "k_chol_lookahead"], m**3 / 3.0 + 2.0 * n * m * m),
        "Sigma -= K T^T (k_syrk_sub)": (["k_syrk_sub"], 2.0 * n * n * m),
        "T = Sigma C^T, S = C T + R (k_build_Z)": (["k_build_Z"], 2.0 * n * n * m + 2.0 * n * m * m),
        "propagate F Sigma F^T (k_propagate_main | k_gemm_nt)": (["k_propagate_main", "k_gemm_nt"], flops_propagate(n)),
    }
    fam_time = {f: sum(per_frame.get(kn, 0.0) for kn in kns) for f, (kns, _) in fam.items()}
    dom = max(fam_time, key=fam_time.get)
    dom_us_all_timed = fam_time[dom]
    dom_us = dom_us_all_timed
    if dom_only and any(kn in dom_only for kn in fam[dom][0]):
        dom_us = sum(dom_only.get(kn, 0.0) for kn in fam[dom][0])
    dom_launches = sum(launches.get(kn, 0.0) for kn in fam[dom][0])
    achieved = fam[dom][1] / (dom_us * 1e-6) / 1e12
    sclk = C.c_double()
    lib.eqf_mfma_f64_peak_clock(core, C.byref(tpeak), C.byref(sclk))
    return {
        "bound": "mfma",
        "kernel": dom,
        "achieved": achieved,
        "peak": FP64_MFMA_PEAK_TFLOPS,
        "unit": "TFLOP/s",
        "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
        "traffic": None,  # HBM-side bytes per launch of the dominant kernel: filled in by live_pmc() from a rocprofv3 counter pass of this very build
        "dominant_kernels": [k_ for k_ in fam[dom][0] if launches.get(k_, 0) > 0],
        "launches_per_frame_by_kernel": {k_: round(v, 3) for k_, v in launches.items()},
        "launches_per_frame": dom_launches,
        "avg_launch_us": dom_us / max(dom_launches, 1.0),
        "avg_launch_us_with_every_kernel_bracketed": dom_us_all_timed / max(dom_launches, 1.0),
        "algorithmic_flops_per_launch": fam[dom][1] / max(dom_launches, 1.0),
        # round 4: with the output blocks evaluated by the propagation kernel there is no k_build_Z launch, the look-ahead kernel builds Z = [S; T; y^T] in front of its
        # first panel (~6 us of its span). `achieved` / `frac` keep counting the factorisation's flops only (conservative, comparable with earlier rounds); counting the
        # dense-formulation flops of T = Sigma C^T, S = C T + R (2 n^2 m + 2 n m^2, of which the kernel executes a fraction: C is 2 x 3 block sparse) as well would give:
        "dominant_kernel_also_builds_Z": bool(launches.get("k_build_Z", 0.0) == 0.0 and launches.get("k_chol_lookahead", 0.0) > 0.0),
        "frac_if_Z_building_counted": ((fam[dom][1] + 2.0 * n * n * m + 2.0 * n * m * m) / (dom_us * 1e-6) / 1e12 / FP64_MFMA_PEAK_TFLOPS)
        if (launches.get("k_build_Z", 0.0) == 0.0 and launches.get("k_chol_lookahead", 0.0) > 0.0) else None,
        "measured_mfma_f64_issue_ceiling_tflops": tpeak.value,
        "sclk_ghz_during_mfma_ceiling": round(sclk.value, 3),
        "mfma_ceiling_note": "k_mfma_peak (8 waves per SIMD, 4 independent accumulators each, no memory traffic) while reading the device's own cycle counter against its 100 MHz "
                             "wall clock: the datasheet's 78.6 TFLOP/s = 1024 SIMDs x 32 flop/clock x 2.4 GHz; at the clock held under this load the same issue rate gives "
                             "%.1f TFLOP/s, i.e. the measured ceiling is %.0f %% of full issue at that clock (one wave alone issues v_mfma_f64_16x16x4_f64 every 64 cycles = full rate: "
                             "scripts/ubench/issue.hip)" % (32768 * sclk.value / 1e3, 100.0 * tpeak.value / max(32768 * sclk.value / 1e3, 1e-9)),
        "per_kernel_us_per_frame": {k_: round(v, 2) for k_, v in sorted(per_frame.items(), key=lambda kv: -kv[1])},
        "note": "hipEvent spans on the filter's own stream over %d frames of the same workload right after the timed region, one span per launch (the launch chain k_chol_step, when selected, is ONE span over its back-to-back launches divided by their number); achieved / frac / avg_launch_us from a second pass of %d frames with only the dominant kernel's launches bracketed (the frame around it as in the timed region), per_kernel_us_per_frame from the pass with every kernel bracketed" % (k, k),
        # round 4: in the TIMED region k_lift and k_syrk_sub are one launch (k_syrk_lift, EQF_OPT_LIFT_WITH_SYRK: the rocprofv3 summary under profiles/ shows it); the span pass
        # above launches them apart, which is what per-kernel spans need
        "timed_region_launches_per_frame": (sum(launches.values()) - 1.0) if (lift_with_syrk and launches.get("k_lift", 0.0) > 0.0 and launches.get("k_syrk_sub", 0.0) > 0.0) else sum(launches.values()),
        "timed_region_note": "k_lift + k_syrk_sub run as ONE launch (k_syrk_lift) in the timed region; the span pass launches them apart" if lift_with_syrk else None,
    }


def _pmc_pass(N, counters, steps=40, warmup=10):
    """One rocprofv3 counter pass over a short run of this script (same build, same workload): {kernel family: {counter: (dispatches, average per dispatch)}}.
    Counters only (--kernel-trace to attribute them to dispatches, no other trace domain), from /tmp as the guide prescribes."""
    import glob
    import re
    import shutil
    import sqlite3
    import subprocess
    import tempfile

    tmp = tempfile.mkdtemp(prefix="eqvio_pmc_", dir="/tmp")
    try:
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", *counters, "-d", tmp, "-o", "p", "--", sys.executable, os.path.join(ROOT, "bench.py"), "--landmarks", str(N), "--steps", str(steps),
               "--warmup", str(warmup), "--no-cpu-baseline", "--no-roofline", "--no-multi-filter", "--no-frame-mix", "--no-binding", "--no-sizes", "--no-pmc"]
        subprocess.run(cmd, cwd="/tmp", env={**os.environ, "TMPDIR": "/tmp"}, capture_output=True, timeout=600, check=True)
        dbs = glob.glob(os.path.join(tmp, "**", "*.db"), recursive=True)
        if not dbs:
            raise RuntimeError("rocprofv3 wrote no database")
        cur = sqlite3.connect(dbs[0]).cursor()
        # a counter has one row per hardware instance and dispatch (32 SQ instances, 8 GRBM): SQ counters are SUMMED over their instances, GRBM_GUI_ACTIVE (the
        # same interval seen by every instance) is averaged; derived counters (FETCH_SIZE, WRITE_SIZE) come as one row per dispatch
        rows = cur.execute("""select s.kernel_name, p.name, count(distinct e.event_id), sum(e.value), avg(e.value) from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
                              join rocpd_kernel_dispatch d on e.event_id = d.event_id join rocpd_info_kernel_symbol s on d.kernel_id = s.id
                              group by s.kernel_name, p.name""").fetchall()
        out = {}
        for name, counter, cnt, total, mean in rows:
            avg = mean if counter.startswith("GRBM") else total / cnt
            mm = re.search(r"(k_[a-z_A-Z0-9]+?)(?:<|I[A-Za-z0-9_]*E|\(|$)", name.replace("eqf::", ""))
            if not mm:
                continue
            fam = re.sub(r"I[a-zA-Z]?L?[bi]?\d.*$", "", mm.group(1))
            d, a = out.setdefault(fam, {}).get(counter, (0, 0.0))
            out[fam][counter] = (d + cnt, (a * d + avg * cnt) / (d + cnt))  # template instantiations of one kernel: averaged by dispatch count
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def live_pmc(N, roofline, updates_per_s):
    """The counter half of the roofline object, measured on THIS build in THIS run: three rocprofv3 passes (FETCH_SIZE / WRITE_SIZE cannot share a pass:
    /opt/skills/guides/MI355X_MICROARCH.md, PMC slots) over 50 frames of the same workload. FETCH_SIZE / WRITE_SIZE arrive in KiB per dispatch. The guide's gfx950
    correction - FETCH_SIZE reports half the bytes of a wide (16 B per lane) coalesced read - is given as a second figure: these kernels read 8 B per lane (fp64
    operands in MFMA fragment layout), for which the counter is uncalibrated, so the truth lies between the two."""
    import shutil

    if shutil.which("rocprofv3") is None:
        return {"pmc_note": "rocprofv3 not on PATH: no counter pass"}
    try:
        fetch = _pmc_pass(N, ["FETCH_SIZE"])
        write = _pmc_pass(N, ["WRITE_SIZE"])
        busy = _pmc_pass(N, ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"])
    except Exception as e:  # informational: never lose the bench line over the profiler
        return {"pmc_note": "counter pass failed: " + repr(e)[:300]}
    lpf = dict(roofline["launches_per_frame_by_kernel"])
    if any("k_syrk_lift" in t for t in (fetch, write, busy)):  # the counter passes run the timed region's form: lift and covariance update as one launch
        lpf["k_syrk_lift"] = lpf.pop("k_syrk_sub", 0.0)
        lpf.pop("k_lift", None)
    dom = roofline["dominant_kernels"]

    def per_launch(table, counter, names):
        tot = sum(lpf.get(k, 0.0) for k in names if k in table and counter in table[k])
        return sum(table[k][counter][1] * lpf.get(k, 0.0) for k in names if k in table and counter in table[k]) / tot if tot > 0 else None

    f_dom, w_dom = per_launch(fetch, "FETCH_SIZE", dom), per_launch(write, "WRITE_SIZE", dom)
    frame_f = sum(fetch[k]["FETCH_SIZE"][1] * lpf[k] for k in lpf if k in fetch and "FETCH_SIZE" in fetch[k]) * 1024.0
    frame_w = sum(write[k]["WRITE_SIZE"][1] * lpf[k] for k in lpf if k in write and "WRITE_SIZE" in write[k]) * 1024.0
    out = {}
    if f_dom is not None and w_dom is not None:
        out["traffic"] = 1024.0 * (f_dom + w_dom)
        out["traffic_fetch_x2"] = 1024.0 * (2.0 * f_dom + w_dom)
    out["hbm_gbps"] = (frame_f + frame_w) * updates_per_s / 1e9
    out["hbm_gbps_fetch_x2"] = (2.0 * frame_f + frame_w) * updates_per_s / 1e9
    out["hbm_frac_of_8tbps"] = out["hbm_gbps_fetch_x2"] / 8000.0
    out["bytes_per_frame_by_kernel"] = {k: round(1024.0 * (fetch.get(k, {}).get("FETCH_SIZE", (0, 0.0))[1] + write.get(k, {}).get("WRITE_SIZE", (0, 0.0))[1]) * lpf[k]) for k in lpf
                                        if k in fetch or k in write}
    mb = {}
    for k in lpf:
        if k in busy and "SQ_VALU_MFMA_BUSY_CYCLES" in busy[k] and busy[k].get("GRBM_GUI_ACTIVE", (0, 0.0))[1] > 0:
            mb[k] = round(busy[k]["SQ_VALU_MFMA_BUSY_CYCLES"][1] / (busy[k]["GRBM_GUI_ACTIVE"][1] * 1024.0), 4)  # busy SIMD-cycles over 1024 SIMDs x the kernel's cycles
    out["mfma_busy_frac_by_kernel"] = mb
    out["mfma_busy_frac"] = next((mb[k] for k in dom if k in mb), None)
    out["pmc_note"] = ("rocprofv3 --kernel-trace --pmc, three passes of 50 frames of this workload on this build (FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE); "
                       "traffic = (FETCH_SIZE + WRITE_SIZE) per launch of the dominant kernel as reported, traffic_fetch_x2 with the guide's gfx950 correction for wide reads "
                       "(these kernels load 8 B per lane: uncalibrated, the truth lies between); hbm_gbps = bytes of every kernel of a frame x the measured frame rate; "
                       "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES summed over the 32 SQ instances / (GRBM_GUI_ACTIVE x 1024 SIMDs) of the dominant kernel")
    return out


if __name__ == "__main__":
    if len(sys.argv) == 6 and sys.argv[1] == "--cpu-batch-worker":
        _cpu_batch_worker(*(int(v) for v in sys.argv[2:]))
    else:
        main()
