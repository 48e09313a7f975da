// Driver for tests/test_integration_filter.py and bench.py: a caller shaped like src/main_sim.cpp:128-184 (the loop over IMU and vision measurements, the
// state estimate read after every frame, Sigma through viewEqFState()) over the reference-side binding VIOFilter_mi355x.cpp + VIO_eqf_mi355x.cpp.
// usage: run_filter_frames <scenario.bin> <out.bin> <fused 0|1> <dump state every k frames, 0 = never> <dump Sigma every k frames, 0 = never> [warm-up frames]
// Prints "frames F seconds S updates_per_s U" (time spent inside processIMUData / processVisionData only).
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "eqvio/VIOFilter.h"

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
    VIOFilter filter(xi0, s, rd<double>());
    // ---- the measurements, read up front (a data server's job; not timed)
    const int nFrames = rd<int32_t>();
    std::vector<std::vector<IMUVelocity>> imus(nFrames);
    std::vector<VisionMeasurement> vision(nFrames);
    for (int f = 0; f < nFrames; ++f) {
        const int k = rd<int32_t>();
        imus[f].resize(k);
        for (int i = 0; i < k; ++i) {
            double v[13];
            rdv(v, 13);
            IMUVelocity& u = imus[f][i];
            u.stamp = v[0];
            u.gyr = Eigen::Vector3d(v[1], v[2], v[3]), u.acc = Eigen::Vector3d(v[4], v[5], v[6]);
            u.gyrBiasVel = Eigen::Vector3d(v[7], v[8], v[9]), u.accBiasVel = Eigen::Vector3d(v[10], v[11], v[12]);
        }
        vision[f].stamp = rd<double>();
        vision[f].cameraPtr = camPtr;
        const int M = rd<int32_t>();
        std::vector<int32_t> ids(M);
        for (int j = 0; j < M; ++j)
            ids[j] = rd<int32_t>();
        for (int j = 0; j < M; ++j) {
            double y[2];
            rdv(y, 2);
            vision[f].camCoordinates[ids[j]] = Eigen::Vector2d(y[0], y[1]);
        }
    }
    // ---- src/main_sim.cpp:128-184
    double seconds = 0.0;
    int visionDataCounter = 0;
    for (int f = 0; f < nFrames; ++f) {
        const auto t0 = std::chrono::steady_clock::now();
        for (const IMUVelocity& imuData : imus[f])
            filter.processIMUData(imuData);
        filter.processVisionData(vision[f]);
        const VIOState estimatedState = filter.stateEstimate(); // every frame, as main_sim does (:148) - one more device round trip, inside the timed region
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
            const VIO_eqf& view = filter.viewEqFState(); // pull(): the writers' access path (src/VIOWriter.cpp:162-222)
            const int32_t rec[3] = {2, f, (int32_t)view.Sigma.rows()};
            std::fwrite(rec, sizeof(int32_t), 3, fout);
            std::fwrite(view.Sigma.data(), sizeof(double), (size_t)view.Sigma.rows() * view.Sigma.cols(), fout);
        }
    }
    std::fclose(fout);
    std::printf("frames %d seconds %.9f updates_per_s %.3f\n", visionDataCounter, seconds, visionDataCounter / seconds);
    return 0;
}
