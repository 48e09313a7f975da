"""SURVEY.md §8(d) measurement configs 1, 2, 3, 5 on one MI355X (config 4 is `bench.py --gpus N`).
Each simulated config drives the device filter main_sim-style from the C++ SimulationDataServer; a bounded prefix of the
same run goes through the CPU oracle in lockstep for parity and the CPU time. Prints one JSON object.
It lives under tests/ because it uses the oracle (as the checker), which only tests/, smoke() and bench.py's cpu_baseline may do."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from eqvio_amd.capi import COORD_EUCLIDEAN, COORD_INVDEPTH, EqfCore, Settings, SimSettings, SimulationDataServer, VIOFilter
from oracle_binding import OracleFilter, se3_log_dist
from util import rel_fro


from eqvio_amd.configs import euroc_settings, sim_consistent, template_settings, uzhfpv_settings  # noqa: E402,F401


def parity(flt, orc):
    s_g, ids_g, p_g = flt.state_estimate()
    s_o, ids_o, p_o = orc.state_estimate()
    assert np.array_equal(ids_g, ids_o), "landmark bookkeeping decisions diverged"
    e = max(se3_log_dist(s_g[6:13], s_o[6:13]) / max(1.0, np.linalg.norm(s_o[10:13])), se3_log_dist(s_g[16:23], s_o[16:23]), np.max(np.abs(s_g[13:16] - s_o[13:16])),
            np.max(np.abs(s_g[0:6] - s_o[0:6])))
    if len(p_o):
        e = max(e, np.max(np.linalg.norm(p_g - p_o, axis=1) / np.maximum(1.0, np.linalg.norm(p_o, axis=1))))
    return e, rel_fro(flt.get_sigma(), orc.get_sigma())


def run_sim(name, fs, sim_kw, augment, oracle_frames):
    sim = SimSettings.defaults(**sim_kw)
    srv = SimulationDataServer(sim, fs)
    fs.cameraOffset[:] = srv.camera_offset()
    s0, ids0, p0 = srv.true_state(0.0, True)
    if not augment:  # start from the features of the first frame only (main_opt-like: the filter adds landmarks itself)
        ids0, p0 = ids0[:0], p0[:0]
    flt = VIOFilter(fs, max_landmarks=max(sim.numPoints + sim.maxFeatures, 64) if augment else 2 * sim.maxFeatures + 64, sensor=s0, ids=ids0, p=p0, time=0.0)
    for tok in os.environ.get("EQF_OPTS", "").split(","):  # e.g. EQF_OPTS=7:0,6:0 to switch device options for a diagnosis run
        if ":" in tok:
            flt.set_core_option(int(tok.split(":")[0]), int(tok.split(":")[1]))
    orc = OracleFilter(fs, s0, ids0, p0, 0.0) if oracle_frames else None
    # a second oracle in the other dense arithmetic (the first is "as written": LU inverse, K evaluated twice; the second "efficient dense": Cholesky): its distance from the
    # first is the rounding floor this configuration's conditioning allows ANY two fp64 implementations
    orc2 = OracleFilter(fs, s0, ids0, p0, 0.0) if oracle_frames else None
    if orc2:
        from oracle_binding import ARITH_AS_WRITTEN, ARITH_EFFICIENT
        orc.set_arithmetic(ARITH_AS_WRITTEN)
        orc2.set_arithmetic(ARITH_EFFICIENT)
    floor_state = floor_sigma = 0.0
    diverged_at = None
    frames = n_lm = 0
    t_dev = t_orc = 0.0
    worst_state = worst_sigma = 0.0
    nees = []
    nees_failures = 0
    while srv.next_measurement_type() != srv.NONE:
        if srv.next_measurement_type() == srv.IMU:
            imu = srv.get_imu()
            flt.process_imu(imu)
            if orc and frames < oracle_frames:
                orc.process_imu(imu)
                orc2.process_imu(imu)
            continue
        stamp, ids, y = srv.get_vision()
        if augment:
            ts, tids, tp = srv.true_state(stamp, True)
        t0 = time.perf_counter()
        if augment:
            flt.augment_landmark_states(ids, ts, tids, tp)
        flt.process_vision(stamp, srv.cam, ids, y)
        flt.synchronize()
        t_dev += time.perf_counter() - t0
        if orc and frames < oracle_frames:
            t0 = time.perf_counter()
            if augment:
                orc.augment_landmark_states(ids, ts, tids, tp)
            orc.process_vision(stamp, srv.cam, ids, y)
            t_orc += time.perf_counter() - t0
            if augment:
                orc2.augment_landmark_states(ids, ts, tids, tp)
            orc2.process_vision(stamp, srv.cam, ids, y)
            if diverged_at is None:
                try:
                    es, eS = parity(flt, orc)
                    fs_, fS_ = parity(orc2, orc)
                    worst_state, worst_sigma = max(worst_state, es), max(worst_sigma, eS)
                    floor_state, floor_sigma = max(floor_state, fs_), max(floor_sigma, fS_)
                except AssertionError:
                    diverged_at = frames  # an outlier / bookkeeping decision flipped (device or second oracle)
        frames += 1
        n_lm += flt.sigma_dim()
        if frames % 10 == 0:
            ts2, tids2, tp2 = srv.true_state(flt.get_time())
            try:
                nees.append(flt.compute_nees(ts2, tids2, tp2))
            except RuntimeError:
                # the device factorises Sigma (Cholesky-type): with the template's 0.003 px noise Sigma reaches cond 5e13 and is
                # SPD only up to rounding; the reference's LU inverse does not notice. Reported, not hidden.
                nees_failures += 1
    est = flt.state_estimate()[0]
    tru = srv.true_state(flt.get_time())[0]
    out = {"frames": frames, "mean_state_dim": n_lm / max(frames, 1), "device_updates_per_s": frames / t_dev, "final_position_error_m": float(np.linalg.norm(est[10:13] - tru[10:13])),
           "mean_nees": float(np.mean(nees)) if nees else None, "nees_not_spd": nees_failures}
    if orc:
        k = min(frames, oracle_frames)
        out.update({"oracle_frames": k, "oracle_updates_per_s": k / t_orc, "parity_state_max": worst_state, "parity_sigma_rel_fro_max": worst_sigma,
                    "oracle_vs_oracle_state_max": floor_state, "oracle_vs_oracle_sigma_rel_fro_max": floor_sigma, "decision_flip_at_frame": diverged_at})
    print(name, json.dumps(out), flush=True)
    return out


def stress500(chart):
    """Config 3: synthetic 500-landmark state, teacher forced against the oracle for 2 frames, device timing over 300 frames after one second of warm-up."""
    from util import euroc_camera, random_imu, reasonable_state, settings_for, synth_measurement
    rng = np.random.default_rng(42)
    N = 500
    settings = settings_for(chart, fastRiccati=1, useDiscreteInnovationLift=0, initialPointVariance=9.0, measurementNoise=1.0)
    xi0, Xs, ids, q0, Q = reasonable_state(rng, N)
    core = EqfCore(N, chart)
    core.set_state(xi0, Xs, ids, q0, Q)
    S0 = np.diag(settings.initial_cov_diag(N))
    core.set_sigma(S0)
    orc = OracleFilter(settings)
    orc.set_eqf(xi0, Xs, ids, q0, Q, S0)
    cam = euroc_camera()
    Qd, Pd = settings.input_gain_diag12(), settings.state_gain_diag8()
    worst = 0.0
    t_orc = 0.0
    for f in range(2):
        imu = random_imu(rng) * np.array([1] + [0.02] * 3 + [0.1] * 3 + [0] * 6)
        _, Xs_, ids_, q0_, Q_ = core.get_state()
        mid, y = synth_measurement(rng, cam, ids_, q0_, Q_, noise_px=1.0)
        core.integrate_riccati_fast(imu, 0.05, Qd, Pd)
        core.vision_update(cam, mid, y, settings.measurementNoise**2, True, False)
        t0 = time.perf_counter()
        orc.integrate_riccati_fast(imu, 0.05)
        orc.vision_update(cam, mid, y)
        t_orc += time.perf_counter() - t0
        worst = max(worst, rel_fro(core.get_sigma(), orc.get_sigma()))
    core.synchronize()
    reps = 300
    imus = [random_imu(rng) * np.array([1] + [0.02] * 3 + [0.1] * 3 + [0] * 6) for _ in range(reps + 20)]  # not inside the timed loop
    t_warm = time.perf_counter()
    while time.perf_counter() - t_warm < 1.0:  # warm-up by the clock: the GPU idled for seconds while the oracle worked on its two frames and has clocked down
        for f in range(reps, reps + 20):
            core.integrate_riccati_fast(imus[f], 0.05, Qd, Pd)
            core.vision_update(cam, mid, y, settings.measurementNoise**2, True, False)
        core.synchronize()
    core.synchronize()
    t0 = time.perf_counter()
    for f in range(reps):  # Sigma keeps evolving
        core.integrate_riccati_fast(imus[f], 0.05, Qd, Pd)
        core.vision_update(cam, mid, y, settings.measurementNoise**2, True, False)
    core.synchronize()
    dt = (time.perf_counter() - t0) / reps
    n, m = 21 + 3 * N, 2 * N
    flops = 4 * n**3 + 24 * n * n + 288 * n + 4 * n * n * m + 4 * n * m * m + m**3 / 3 + 2 * n * m
    out = {"N": N, "state_dim": n, "device_updates_per_s": 1 / dt, "dense_equiv_tflops": flops / dt / 1e12, "oracle_updates_per_s": 2 / t_orc, "parity_sigma_rel_fro_max": worst}
    print("config3", chart, json.dumps(out), flush=True)
    return out


if __name__ == "__main__":
    which = sys.argv[1].split(",") if len(sys.argv) > 1 else ["1", "2", "3", "5"]
    res = {}
    if "1" in which:
        kw = dict(duration=20.0, trajectory="wave", numPoints=1000, wallDistance=2.0, numWalls=1, randomSeed=0, maxFeatures=20)
        res["config1_template_accurate_riccati"] = run_sim("config1/accurate", template_settings(0), kw, True, 60)
        res["config1_template_fast_riccati"] = run_sim("config1/fast", template_settings(1), kw, True, 400)
    if "2" in which:
        kw = dict(duration=144.0, trajectory="sine", numPoints=4000, wallDistance=3.0, numWalls=6, randomSeed=1, maxFeatures=50, outputNoise=1, inputNoise=0)
        res["config2_euroc_standin_50"] = run_sim("config2", sim_consistent(euroc_settings(), measurementNoise=1.0), kw, False, 200)
    if "3" in which:
        res["config3_stress500_invdepth"] = stress500(COORD_INVDEPTH)
        res["config3_stress500_euclid"] = stress500(COORD_EUCLIDEAN)
    if "5" in which:
        kw = dict(duration=30.0, trajectory="sine", numPoints=12000, wallDistance=3.0, numWalls=6, randomSeed=5, maxFeatures=200, imuFreq=500.0, imageFreq=30.0, outputNoise=1, inputNoise=0)
        res["config5_uzhfpv_standin_200_fp64"] = run_sim("config5", sim_consistent(uzhfpv_settings(), measurementNoise=1.0), kw, False, 12)
    print(json.dumps(res))
