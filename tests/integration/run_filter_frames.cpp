// Driver for tests/test_integration_filter.py and bench.py over the reference-side binding VIO_eqf_mi355x.cpp: it REPLAYS, frame by frame, the sequence of
// VIO_eqf member calls that the reference's VIOFilter::processVisionData makes (src/VIOFilter.cpp:194-241 -> :134-192, :280-302, :304-364, :258-278) from a
// PLAN in the scenario file - the clipped interval of every buffered IMU sample, the landmarks the reference drops as lost, the ones its removeOutliers
// rejects, the new landmarks with their initial points. The plan is computed by the test (tests/integration_scenario.py, from the oracle's filter where there
// is something to decide), so that none of the reference's control flow is restated here: this file only makes the calls, in the reference's order.
//   fused = 0: member for member - integrateRiccatiStateFast + k x integrateObserverState, removeLandmarkById (lost), stateEstimate + one getOutputCovById per
//              measured landmark (what removeOutliers asks the filter for, :304-334), removeLandmarkById (outliers), addNewLandmarks, performVisionUpdate,
//              removeInvalidLandmarks.
//   fused = 1: the two hunks of VIOFilter_mi355x_hunks.hpp (stageMeasurement + propagateFast; statsThenUpdate), falling back to the planned calls where the
//              device leaves the outlier decision to the host.
// The state estimate is read after every frame (src/main_sim.cpp:148), Sigma through pull() (the writers' path, src/VIOWriter.cpp:162-222).
// usage: run_filter_frames <scenario.bin> <out.bin> <fused 0|1> <dump state every k frames, 0 = never> <dump Sigma every k frames, 0 = never> [warm-up frames]
// Prints "frames F seconds S updates_per_s U" (time spent in the member calls and the per-frame state read only).
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "VIOFilter_mi355x_hunks.hpp"

const EqFCoordinateSuite EqFCoordinateSuite_euclid{}, EqFCoordinateSuite_invdepth{}, EqFCoordinateSuite_normal{}; // the reference defines these in coordinateSuite/*.cpp

static FILE* fin;
template <typename T> static T rd() {
    T v;
    if (std::fread(&v, sizeof(T), 1, fin) != 1) {
        std::fprintf(stderr, "scenario file truncated\n");
        std::exit(2);
    }
    return v;
}
static void rdv(double* p, size_t n) {
    if (std::fread(p, sizeof(double), n, fin) != n) {
        std::fprintf(stderr, "scenario file truncated\n");
        std::exit(2);
    }
}
static liepp::SE3d pose7(const double* q) { return liepp::SE3d(liepp::SO3d(Eigen::Quaterniond(q[0], q[1], q[2], q[3])), Eigen::Vector3d(q[4], q[5], q[6])); }

int main(int argc, char** argv) {
    if (argc != 6 && argc != 7) {
        std::fprintf(stderr, "usage: run_filter_frames <scenario.bin> <out.bin> <fused> <state every> <Sigma every> [untimed warm-up frames]\n");
        return 2;
    }
    const int warm = argc == 7 ? std::atoi(argv[6]) : 0; // the first frames create the device context (~0.1 s): not part of a steady-state rate
    fin = std::fopen(argv[1], "rb");
    FILE* fout = std::fopen(argv[2], "wb");
    if (!fin || !fout)
        return 2;
    const int fused = std::atoi(argv[3]), stateEvery = std::atoi(argv[4]), sigmaEvery = std::atoi(argv[5]);
    // ---- settings (the eqf block), in the order of include/eqvio_types.h eqvio_settings: 26 doubles, 9 ints, cameraOffset
    VIOFilter::Settings s;
    double* sd[26] = {&s.biasOmegaProcessVariance, &s.biasAccelProcessVariance, &s.attitudeProcessVariance, &s.positionProcessVariance, &s.velocityProcessVariance,
                      &s.cameraAttitudeProcessVariance, &s.cameraPositionProcessVariance, &s.pointProcessVariance, &s.velGyrNoise, &s.velAccNoise, &s.velGyrBiasWalk,
                      &s.velAccBiasWalk, &s.measurementNoise, &s.outlierThresholdAbs, &s.outlierThresholdProb, &s.featureRetention, &s.initialAttitudeVariance,
                      &s.initialPositionVariance, &s.initialVelocityVariance, &s.initialCameraAttitudeVariance, &s.initialCameraPositionVariance, &s.initialPointVariance,
                      &s.initialPointDepthVariance, &s.initialBiasOmegaVariance, &s.initialBiasAccelVariance, &s.initialSceneDepth};
    for (double* p : sd)
        *p = rd<double>();
    s.useDiscreteInnovationLift = rd<int32_t>(), s.useDiscreteVelocityLift = rd<int32_t>(), s.useDiscreteStateMatrix = rd<int32_t>(), s.fastRiccati = rd<int32_t>();
    s.useMedianDepth = rd<int32_t>(), s.useFeaturePredictions = rd<int32_t>(), s.useEquivariantOutput = rd<int32_t>(), s.removeLostLandmarks = rd<int32_t>();
    const int chart = rd<int32_t>();
    s.coordinateChoice = chart == 0 ? CoordinateChoice::Euclidean : chart == 1 ? CoordinateChoice::InvDepth : CoordinateChoice::Normal;
    double off[7];
    rdv(off, 7);
    s.cameraOffset = pose7(off);
    s.mi355xFused = fused != 0;
    // ---- camera, initial condition (src/main_sim.cpp:103: VIOFilter filter(simDataServer.getInitialCondition(), filterSettings))
    double cam[6];
    rdv(cam, 6);
    const GIFT::GICameraPtr camPtr = std::make_shared<GIFT::PinholeCamera>(GIFT::ImageSize{(int)cam[4], (int)cam[5]}, cam[0], cam[1], cam[2], cam[3]);
    double sensor[23];
    rdv(sensor, 23);
    const int N0 = rd<int32_t>();
    VIOState xi0;
    for (int i = 0; i < 6; ++i)
        xi0.sensor.inputBias(i) = sensor[i];
    xi0.sensor.pose = pose7(sensor + 6);
    xi0.sensor.velocity = Eigen::Vector3d(sensor[13], sensor[14], sensor[15]);
    xi0.sensor.cameraOffset = pose7(sensor + 16);
    xi0.cameraLandmarks.resize(N0);
    for (int i = 0; i < N0; ++i)
        xi0.cameraLandmarks[i].id = rd<int32_t>();
    for (int i = 0; i < N0; ++i) {
        double p[3];
        rdv(p, 3);
        xi0.cameraLandmarks[i].p = Eigen::Vector3d(p[0], p[1], p[2]);
    }
    VIO_eqf filterState; // the construction of src/VIOFilter.cpp:43-56
    filterState.Sigma = s.constructInitialStateCovariance(xi0.cameraLandmarks.size());
    filterState.xi0 = xi0;
    for (const Landmark& lm : xi0.cameraLandmarks) {
        filterState.X.Q.emplace_back(liepp::SOT3d::Identity());
        filterState.X.id.emplace_back(lm.id);
    }
    filterState.coordinateSuite = getCoordinates(s.coordinateChoice);
    filterState.currentTime = rd<double>();
    filterState.markHostEdited(); // hook of INTEGRATION.md section A: xi0 / X / Sigma were assigned directly
    // ---- the measurements and the plan, read up front (a data server's job; not timed)
    struct Frame {
        std::vector<IMUVelocity> samples; // the buffered samples that overlap the frame's interval ...
        std::vector<double> dt;           // ... and their clipped intervals
        VisionMeasurement vision;
        std::vector<int> lost, outliers;  // landmark ids, in the order the reference removes them
        std::vector<Landmark> fresh;      // new landmarks, initial points included
    };
    const int nFrames = rd<int32_t>();
    std::vector<Frame> plan(nFrames);
    for (Frame& fr : plan) {
        const int k = rd<int32_t>();
        fr.samples.resize(k), fr.dt.resize(k);
        for (int i = 0; i < k; ++i) {
            double v[14];
            rdv(v, 14);
            IMUVelocity& u = fr.samples[i];
            u.stamp = v[0];
            u.gyr = Eigen::Vector3d(v[1], v[2], v[3]), u.acc = Eigen::Vector3d(v[4], v[5], v[6]);
            u.gyrBiasVel = Eigen::Vector3d(v[7], v[8], v[9]), u.accBiasVel = Eigen::Vector3d(v[10], v[11], v[12]);
            fr.dt[i] = v[13];
        }
        fr.vision.stamp = rd<double>();
        fr.vision.cameraPtr = camPtr;
        const int M = rd<int32_t>();
        std::vector<int32_t> ids(M);
        for (int32_t& id : ids)
            id = rd<int32_t>();
        for (int j = 0; j < M; ++j) {
            double y[2];
            rdv(y, 2);
            fr.vision.camCoordinates[ids[j]] = Eigen::Vector2d(y[0], y[1]);
        }
        fr.lost.resize(rd<int32_t>());
        for (int& id : fr.lost)
            id = rd<int32_t>();
        fr.outliers.resize(rd<int32_t>());
        for (int& id : fr.outliers)
            id = rd<int32_t>();
        fr.fresh.resize(rd<int32_t>());
        for (Landmark& lm : fr.fresh)
            lm.id = rd<int32_t>();
        for (Landmark& lm : fr.fresh) {
            double p[3];
            rdv(p, 3);
            lm.p = Eigen::Vector3d(p[0], p[1], p[2]);
        }
    }
    double seconds = 0.0, gainSeconds = 0.0; // gainSeconds: inside the reference's own dense gain-matrix constructors (VIOFilterSettings.h:176-206), caller side
    int visionDataCounter = 0;
    bool timedFrame = false;
    const auto timed = [&](auto&& make) { // evaluates make() and books its time under gainSeconds
        const auto g0 = std::chrono::steady_clock::now();
        auto m = make();
        if (timedFrame)
            gainSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - g0).count();
        return m;
    };
    const auto add_fresh = [&](const Frame& fr) { // VIO_eqf::addNewLandmarks as src/VIOFilter.cpp:273-277 calls it
        if (fr.fresh.empty())
            return;
        std::vector<Landmark> lms = fr.fresh;
        const int k3 = 3 * (int)lms.size();
        filterState.addNewLandmarks(lms, Eigen::MatrixXd::Identity(k3, k3) * s.initialPointVariance);
    };
    const auto planned_update = [&](const Frame& fr, bool fresh_already_added) { // the calls behind the outlier decision, :225-240
        VisionMeasurement matched = fr.vision;
        for (const int id : fr.outliers) {
            filterState.removeLandmarkById(id);
            matched.camCoordinates.erase(id);
        }
        if (!fresh_already_added)
            add_fresh(fr);
        if (!matched.camCoordinates.empty())
            filterState.performVisionUpdate(matched, timed([&] { return s.constructOutputGainMatrix(matched.camCoordinates.size()); }), s.useEquivariantOutput, s.useDiscreteInnovationLift);
        filterState.removeInvalidLandmarks();
    };
    for (int f = 0; f < nFrames; ++f) {
        const Frame& fr = plan[f];
        timedFrame = f >= warm;
        const auto t0 = std::chrono::steady_clock::now();
        double totalTime = 0.0;
        IMUVelocity meanVelocity = IMUVelocity::Zero();
        for (size_t i = 0; i < fr.samples.size(); ++i) {
            totalTime += fr.dt[i];
            meanVelocity = meanVelocity + fr.samples[i] * fr.dt[i];
        }
        meanVelocity = meanVelocity * (1.0 / totalTime);
        if (fused) {
            filterState.stageMeasurement(fr.vision); // the measurement travels to HBM inside the propagation kernel
            eqvio_mi355x::fusedPropagation(filterState, s, meanVelocity, totalTime, fr.samples, fr.dt);
        } else {
            filterState.integrateRiccatiStateFast(meanVelocity, totalTime, s.constructInputGainMatrix(), timed([&] { return s.constructStateGainMatrix(filterState.xi0.cameraLandmarks.size()); }));
            for (size_t i = 0; i < fr.samples.size(); ++i)
                filterState.integrateObserverState(fr.samples[i], fr.dt[i], s.useDiscreteVelocityLift);
        }
        filterState.currentTime = fr.vision.stamp;
        for (const int id : fr.lost)
            filterState.removeLandmarkById(id);
        if (fused) {
            const bool early = !s.useMedianDepth; // with a fixed initial depth the new landmarks may enter before the outlier test (it never looks at them)
            if (early)
                add_fresh(fr);
            std::vector<double> absErr, probErr;
            if (eqvio_mi355x::fusedStatsAndUpdate(filterState, s, fr.vision, early, absErr, probErr) == 1)
                filterState.removeInvalidLandmarks();
            else
                planned_update(fr, early);
        } else {
            // what removeOutliers asks the filter for, whatever it then decides: the state estimate and the output covariance of every measured landmark
            const VIOState xiHat = filterState.stateEstimate();
            double sink = 0.0;
            for (const Landmark& lm : xiHat.cameraLandmarks) {
                const auto it = fr.vision.camCoordinates.find(lm.id);
                if (it != fr.vision.camCoordinates.end())
                    sink += filterState.getOutputCovById(lm.id, it->second, camPtr)(0, 0);
            }
            if (sink == 12345.678)
                std::fprintf(stderr, "\n");
            planned_update(fr, false);
        }
        const VIOState estimatedState = filterState.stateEstimate(); // every frame, as main_sim does (:148) - one more device round trip, inside the timed region
        if (f >= warm) {
            seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            ++visionDataCounter;
        }
        if (stateEvery > 0 && f % stateEvery == 0) {
            const int32_t rec[3] = {1, f, (int32_t)estimatedState.cameraLandmarks.size()};
            std::fwrite(rec, sizeof(int32_t), 3, fout);
            double sv[23];
            for (int i = 0; i < 6; ++i)
                sv[i] = estimatedState.sensor.inputBias(i);
            const auto pack = [](const liepp::SE3d& P, double* q) {
                const Eigen::Quaterniond a = P.R.asQuaternion();
                q[0] = a.w(), q[1] = a.x(), q[2] = a.y(), q[3] = a.z(), q[4] = P.x(0), q[5] = P.x(1), q[6] = P.x(2);
            };
            pack(estimatedState.sensor.pose, sv + 6);
            for (int i = 0; i < 3; ++i)
                sv[13 + i] = estimatedState.sensor.velocity(i);
            pack(estimatedState.sensor.cameraOffset, sv + 16);
            std::fwrite(sv, sizeof(double), 23, fout);
            for (const Landmark& lm : estimatedState.cameraLandmarks) {
                const int32_t id = lm.id;
                std::fwrite(&id, sizeof(int32_t), 1, fout);
            }
            for (const Landmark& lm : estimatedState.cameraLandmarks)
                std::fwrite(lm.p.data(), sizeof(double), 3, fout);
        }
        if (sigmaEvery > 0 && f % sigmaEvery == 0) {
            filterState.pull(); // hook of INTEGRATION.md section A: what VIOFilter::viewEqFState() does before it hands the members to the writers
            const VIO_eqf& view = filterState;
            const int32_t rec[3] = {2, f, (int32_t)view.Sigma.rows()};
            std::fwrite(rec, sizeof(int32_t), 3, fout);
            std::fwrite(view.Sigma.data(), sizeof(double), (size_t)view.Sigma.rows() * view.Sigma.cols(), fout);
        }
    }
    std::fclose(fout);
    std::printf("frames %d seconds %.9f updates_per_s %.3f gain_matrix_seconds %.9f\n", visionDataCounter, seconds, visionDataCounter / seconds, gainSeconds);
    return 0;
}
