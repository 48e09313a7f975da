"""BASELINE.json configs[3] (`bench.py --gpus N --config batch8`): the batch of 8 sequences in the form SURVEY.md section 8(d) gives it for a box without the datasets - the C++
SimulationDataServer on {wave, square, sine, line} x maxFeatures {40, 200}, seeds 0-7. Two of the eight sequences against the oracle's filter, frame by frame and teacher
forced (flat 1e-9; the filter adds and drops its landmarks itself, main_opt-like), and the launch of all 8 ranks on ONE device (EQVIO_BENCH_ONE_DEVICE, what a 1-GPU box can
rehearse of the 8-GPU run)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_batch8_is_the_eight_sequences_of_the_survey():
    assert [(s["trajectory"], s["maxFeatures"], s["seed"]) for s in bench.BATCH8] == [("wave", 40, 0), ("square", 40, 1), ("sine", 40, 2), ("line", 40, 3), ("wave", 200, 4),
                                                                                       ("square", 200, 5), ("sine", 200, 6), ("line", 200, 7)]


@pytest.mark.gpu
@pytest.mark.parametrize("q", [1, 6])
def test_batch8_sequence_follows_the_oracle(q):
    from eqvio_amd.capi import VIOFilter
    from oracle_binding import OracleFilter
    from run_configs import parity
    from util import teacher_force

    fs, cam, s0, frames = bench.batch8_sequence(q, duration=4.0)
    none_i, none_p = np.zeros(0, np.int32), np.zeros((0, 3))
    flt = VIOFilter(fs, max_landmarks=2 * bench.BATCH8[q]["maxFeatures"] + 64, sensor=s0, ids=none_i, p=none_p, time=0.0)
    orc = OracleFilter(fs, s0, none_i, none_p, 0.0)
    worst, seen, sizes = 0.0, set(), []
    for imus, stamp, ids, y in frames:
        for u in imus:
            flt.process_imu(u)
            orc.process_imu(u)
        flt.process_vision(stamp, cam, ids, y)
        orc.process_vision(stamp, cam, ids, y)
        es, eS = parity(flt, orc)  # asserts identical landmark sets
        assert es <= 1e-9 and eS <= 1e-9, (stamp, es, eS)
        worst = max(worst, es, eS)
        teacher_force(flt, orc)
        seen |= set(ids.tolist())
        sizes.append((flt.sigma_dim() - 21) // 3)
    want = bench.BATCH8[q]["maxFeatures"]
    assert len(frames) >= 75 and max(sizes) >= 0.9 * want and len(seen) > max(sizes), (len(frames), max(sizes), len(seen))  # the tracked set is full and turns over
    print(f"batch8 sequence {q} ({bench.BATCH8[q]}): {len(frames)} frames teacher forced, worst deviation {worst:.1e}, up to {max(sizes)} landmarks")


@pytest.mark.gpu
def test_batch8_launches_eight_ranks_on_one_device():
    env = dict(os.environ, EQVIO_BENCH_ONE_DEVICE="1", EQVIO_BENCH_SPIN_UP_S="0.05")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--config", "batch8"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and sorted(d["sequences"]) == [str(k) for k in range(8)]
    assert all(v["frames"] == 1200 for v in d["sequences"].values()) and d["steps"] == 9600
    assert d["per_rank_seconds"]["min"] > 0 and d["value"] == pytest.approx(9600 / d["per_rank_seconds"]["max"], rel=1e-6)
    assert all(v["final_landmarks"] >= 0.8 * v["maxFeatures"] for v in d["sequences"].values()), d["sequences"]
