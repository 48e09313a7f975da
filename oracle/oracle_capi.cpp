// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
// C interface (for ctypes) over the CPU restatement in vio.hpp. The entry points mirror the product's
// C-ABI (include/eqvio_filter.h, include/eqf_hip.h) name for name with the prefix orc_ so the parity tests
// drive both through the same harness.
#include "eqvio_types.h"
#include "vio.hpp"
#include <chrono>

using namespace orc;

namespace {
SO3 quat(const double* q) { return SO3::fromQuat(q[0], q[1], q[2], q[3]); }
void putQuat(const SO3& R, double* q) {
    q[0] = R.w;
    q[1] = R.x;
    q[2] = R.y;
    q[3] = R.z;
}
VIOSensorState unpackSensor(const double* s) {
    VIOSensorState r;
    for (int i = 0; i < 6; ++i)
        r.inputBias(i) = s[i];
    r.pose = SE3(quat(s + 6), vec3(s[10], s[11], s[12]));
    r.velocity = vec3(s[13], s[14], s[15]);
    r.cameraOffset = SE3(quat(s + 16), vec3(s[20], s[21], s[22]));
    return r;
}
void packSensor(const VIOSensorState& r, double* s) {
    for (int i = 0; i < 6; ++i)
        s[i] = r.inputBias(i);
    putQuat(r.pose.R, s + 6);
    for (int i = 0; i < 3; ++i) {
        s[10 + i] = r.pose.x(i);
        s[13 + i] = r.velocity(i);
        s[20 + i] = r.cameraOffset.x(i);
    }
    putQuat(r.cameraOffset.R, s + 16);
}
void unpackGroupSensor(const double* s, VIOGroup& X) {
    for (int i = 0; i < 6; ++i)
        X.beta(i) = s[i];
    X.A = SE3(quat(s + 6), vec3(s[10], s[11], s[12]));
    X.w = vec3(s[13], s[14], s[15]);
    X.B = SE3(quat(s + 16), vec3(s[20], s[21], s[22]));
}
void packGroupSensor(const VIOGroup& X, double* s) {
    for (int i = 0; i < 6; ++i)
        s[i] = X.beta(i);
    putQuat(X.A.R, s + 6);
    for (int i = 0; i < 3; ++i) {
        s[10 + i] = X.A.x(i);
        s[13 + i] = X.w(i);
        s[20 + i] = X.B.x(i);
    }
    putQuat(X.B.R, s + 16);
}
IMUVelocity unpackIMU(const double* v) {
    IMUVelocity r;
    r.stamp = v[0];
    r.gyr = vec3(v[1], v[2], v[3]);
    r.acc = vec3(v[4], v[5], v[6]);
    r.gyrBiasVel = vec3(v[7], v[8], v[9]);
    r.accBiasVel = vec3(v[10], v[11], v[12]);
    return r;
}
CameraPtr makeCamera(const eqvio_camera* c) {
    auto cam = std::make_shared<Camera>();
    cam->model = c->model;
    cam->fx = c->fx;
    cam->fy = c->fy;
    cam->cx = c->cx;
    cam->cy = c->cy;
    cam->width = c->width;
    cam->height = c->height;
    for (int i = 0; i < 5; ++i)
        cam->dist[i] = c->dist[i];
    return cam;
}
Settings makeSettings(const eqvio_settings* s) {
    Settings r;
    r.biasOmegaProcessVariance = s->biasOmegaProcessVariance;
    r.biasAccelProcessVariance = s->biasAccelProcessVariance;
    r.attitudeProcessVariance = s->attitudeProcessVariance;
    r.positionProcessVariance = s->positionProcessVariance;
    r.velocityProcessVariance = s->velocityProcessVariance;
    r.cameraAttitudeProcessVariance = s->cameraAttitudeProcessVariance;
    r.cameraPositionProcessVariance = s->cameraPositionProcessVariance;
    r.pointProcessVariance = s->pointProcessVariance;
    r.velGyrNoise = s->velGyrNoise;
    r.velAccNoise = s->velAccNoise;
    r.velGyrBiasWalk = s->velGyrBiasWalk;
    r.velAccBiasWalk = s->velAccBiasWalk;
    r.measurementNoise = s->measurementNoise;
    r.outlierThresholdAbs = s->outlierThresholdAbs;
    r.outlierThresholdProb = s->outlierThresholdProb;
    r.featureRetention = s->featureRetention;
    r.initialAttitudeVariance = s->initialAttitudeVariance;
    r.initialPositionVariance = s->initialPositionVariance;
    r.initialVelocityVariance = s->initialVelocityVariance;
    r.initialCameraAttitudeVariance = s->initialCameraAttitudeVariance;
    r.initialCameraPositionVariance = s->initialCameraPositionVariance;
    r.initialPointVariance = s->initialPointVariance;
    r.initialPointDepthVariance = s->initialPointDepthVariance;
    r.initialBiasOmegaVariance = s->initialBiasOmegaVariance;
    r.initialBiasAccelVariance = s->initialBiasAccelVariance;
    r.initialSceneDepth = s->initialSceneDepth;
    r.useDiscreteInnovationLift = s->useDiscreteInnovationLift;
    r.useDiscreteVelocityLift = s->useDiscreteVelocityLift;
    r.useDiscreteStateMatrix = s->useDiscreteStateMatrix;
    r.fastRiccati = s->fastRiccati;
    r.useMedianDepth = s->useMedianDepth;
    r.useFeaturePredictions = s->useFeaturePredictions;
    r.useEquivariantOutput = s->useEquivariantOutput;
    r.removeLostLandmarks = s->removeLostLandmarks;
    r.coordinateChoice = (CoordinateChoice)s->coordinateChoice;
    r.cameraOffset = SE3(quat(s->cameraOffset), vec3(s->cameraOffset[4], s->cameraOffset[5], s->cameraOffset[6]));
    return r;
}
VisionMeasurement makeMeasurement(double stamp, const eqvio_camera* cam, const int* ids, const double* y, int M) {
    VisionMeasurement m;
    m.stamp = stamp;
    m.cameraPtr = makeCamera(cam);
    for (int i = 0; i < M; ++i)
        m.camCoordinates[ids[i]] = vec2(y[2 * i], y[2 * i + 1]);
    return m;
}
VIOState unpackState(const double* sensor, const int* ids, const double* p, int N) {
    VIOState xi;
    xi.sensor = unpackSensor(sensor);
    xi.cameraLandmarks.resize(N);
    for (int i = 0; i < N; ++i) {
        xi.cameraLandmarks[i].id = ids[i];
        xi.cameraLandmarks[i].p = vec3(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
    }
    return xi;
}
int packState(const VIOState& xi, double* sensor, int* ids, double* p, int cap) {
    const int N = (int)xi.cameraLandmarks.size();
    if (sensor)
        packSensor(xi.sensor, sensor);
    if (N > cap)
        return -N;
    for (int i = 0; i < N; ++i) {
        if (ids)
            ids[i] = xi.cameraLandmarks[i].id;
        if (p)
            for (int k = 0; k < 3; ++k)
                p[3 * i + k] = xi.cameraLandmarks[i].p(k);
    }
    return N;
}
int copyMat(const DMat& Mm, double* out, int cap) {
    if ((int)Mm.d.size() > cap)
        return -(int)Mm.d.size();
    std::memcpy(out, Mm.d.data(), Mm.d.size() * sizeof(double));
    return (int)Mm.d.size();
}
} // namespace

extern "C" {

void orc_default_settings(eqvio_settings* s) {
    const Settings d;
    s->biasOmegaProcessVariance = d.biasOmegaProcessVariance;
    s->biasAccelProcessVariance = d.biasAccelProcessVariance;
    s->attitudeProcessVariance = d.attitudeProcessVariance;
    s->positionProcessVariance = d.positionProcessVariance;
    s->velocityProcessVariance = d.velocityProcessVariance;
    s->cameraAttitudeProcessVariance = d.cameraAttitudeProcessVariance;
    s->cameraPositionProcessVariance = d.cameraPositionProcessVariance;
    s->pointProcessVariance = d.pointProcessVariance;
    s->velGyrNoise = d.velGyrNoise;
    s->velAccNoise = d.velAccNoise;
    s->velGyrBiasWalk = d.velGyrBiasWalk;
    s->velAccBiasWalk = d.velAccBiasWalk;
    s->measurementNoise = d.measurementNoise;
    s->outlierThresholdAbs = d.outlierThresholdAbs;
    s->outlierThresholdProb = d.outlierThresholdProb;
    s->featureRetention = d.featureRetention;
    s->initialAttitudeVariance = d.initialAttitudeVariance;
    s->initialPositionVariance = d.initialPositionVariance;
    s->initialVelocityVariance = d.initialVelocityVariance;
    s->initialCameraAttitudeVariance = d.initialCameraAttitudeVariance;
    s->initialCameraPositionVariance = d.initialCameraPositionVariance;
    s->initialPointVariance = d.initialPointVariance;
    s->initialPointDepthVariance = d.initialPointDepthVariance;
    s->initialBiasOmegaVariance = d.initialBiasOmegaVariance;
    s->initialBiasAccelVariance = d.initialBiasAccelVariance;
    s->initialSceneDepth = d.initialSceneDepth;
    s->useDiscreteInnovationLift = d.useDiscreteInnovationLift;
    s->useDiscreteVelocityLift = d.useDiscreteVelocityLift;
    s->useDiscreteStateMatrix = d.useDiscreteStateMatrix;
    s->fastRiccati = d.fastRiccati;
    s->useMedianDepth = d.useMedianDepth;
    s->useFeaturePredictions = d.useFeaturePredictions;
    s->useEquivariantOutput = d.useEquivariantOutput;
    s->removeLostLandmarks = d.removeLostLandmarks;
    s->coordinateChoice = (int)d.coordinateChoice;
    const double id7[7] = {1, 0, 0, 0, 0, 0, 0};
    std::memcpy(s->cameraOffset, id7, sizeof(id7));
}

// ---- filter lifecycle (VIOFilter ctors, src/VIOFilter.cpp:31-56)
void* orc_filter_create(const eqvio_settings* s) { return new VIOFilter(makeSettings(s)); }
void* orc_filter_create_from_state(const eqvio_settings* s, const double* sensor, const int* ids, const double* p, int N, double time) {
    return new VIOFilter(unpackState(sensor, ids, p, N), makeSettings(s), time);
}
void orc_filter_destroy(void* f) { delete (VIOFilter*)f; }
void orc_filter_set_arithmetic(void* f, int mode) { ((VIOFilter*)f)->filterState.arithmetic = (UpdateArithmetic)mode; }

// ---- VIOFilter public API
void orc_filter_process_imu(void* f, const double* imu13) { ((VIOFilter*)f)->processIMUData(unpackIMU(imu13)); }
void orc_filter_process_vision(void* f, double stamp, const eqvio_camera* cam, const int* ids, const double* y, int M) {
    ((VIOFilter*)f)->processVisionData(makeMeasurement(stamp, cam, ids, y, M));
}
int orc_filter_state_estimate(void* f, double* sensor, int* ids, double* p, int cap) {
    return packState(((VIOFilter*)f)->stateEstimate(), sensor, ids, p, cap);
}
double orc_filter_get_time(void* f) { return ((VIOFilter*)f)->getTime(); }
int orc_filter_is_initialised(void* f) { return ((VIOFilter*)f)->initialisedFlag; }
void orc_filter_set_state(void* f, const double* sensor, const int* ids, const double* p, int N) {
    ((VIOFilter*)f)->setState(unpackState(sensor, ids, p, N));
}
void orc_filter_set_landmarks(void* f, const int* ids, const double* p, int N) {
    const double s0[23] = {0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0};
    ((VIOFilter*)f)->setLandmarks(unpackState(s0, ids, p, N).cameraLandmarks);
}
void orc_filter_augment_landmark_states(void* f, const int* newIds, int nNew, const double* sensor, const int* ids, const double* p, int N) {
    ((VIOFilter*)f)->augmentLandmarkStates(std::vector<int>(newIds, newIds + nNew), unpackState(sensor, ids, p, N));
}

// ---- viewEqFState(): xi0, X, Sigma (VIO_eqf public members, include/eqvio/mathematical/VIO_eqf.h:36-42)
int orc_filter_get_eqf(void* f, double* xi0_sensor, double* X_sensor, int* ids, double* q0, double* Q, int cap) {
    const VIO_eqf& e = ((VIOFilter*)f)->filterState;
    const int N = (int)e.X.id.size();
    packSensor(e.xi0.sensor, xi0_sensor);
    packGroupSensor(e.X, X_sensor);
    if (N > cap)
        return -N;
    for (int i = 0; i < N; ++i) {
        ids[i] = e.X.id[i];
        for (int k = 0; k < 3; ++k)
            q0[3 * i + k] = e.xi0.cameraLandmarks[i].p(k);
        putQuat(e.X.Q[i].R, Q + 5 * i);
        Q[5 * i + 4] = e.X.Q[i].a;
    }
    return N;
}
void orc_filter_set_eqf(void* f, const double* xi0_sensor, const double* X_sensor, const int* ids, const double* q0, const double* Q, int N, const double* Sigma, double time) {
    VIOFilter* vf = (VIOFilter*)f;
    VIO_eqf& e = vf->filterState;
    e.xi0 = unpackState(xi0_sensor, ids, q0, N);
    e.X = VIOGroup::Identity(std::vector<int>(ids, ids + N));
    unpackGroupSensor(X_sensor, e.X);
    for (int i = 0; i < N; ++i)
        e.X.Q[i] = SOT3(quat(Q + 5 * i), Q[5 * i + 4]);
    const int n = 21 + 3 * N;
    e.Sigma = DMat(n, n);
    std::memcpy(e.Sigma.d.data(), Sigma, sizeof(double) * n * n);
    e.currentTime = time;
    vf->initialisedFlag = true;
}
int orc_filter_get_sigma(void* f, double* out, int cap) { return copyMat(((VIOFilter*)f)->filterState.Sigma, out, cap); }
int orc_filter_sigma_dim(void* f) { return ((VIOFilter*)f)->filterState.Sigma.r; }
int orc_filter_last_gamma(void* f, double* out, int cap) {
    const DVec& g = ((VIOFilter*)f)->filterState.lastGamma;
    if ((int)g.size() > cap)
        return -(int)g.size();
    std::memcpy(out, g.data(), g.size() * sizeof(double));
    return (int)g.size();
}

// ---- VIO_eqf methods, callable one by one (src/mathematical/VIO_eqf.cpp)
void orc_eqf_integrate_riccati_fast(void* f, const double* imu13, double dt) {
    VIOFilter* vf = (VIOFilter*)f;
    vf->filterState.integrateRiccatiStateFast(unpackIMU(imu13), dt, vf->settings.inputGainDiag(), vf->settings.stateGainDiag(vf->filterState.X.id.size()));
}
void orc_eqf_integrate_riccati_accurate(void* f, const double* imu13, double dt) {
    VIOFilter* vf = (VIOFilter*)f;
    vf->filterState.integrateRiccatiStateAccurate(unpackIMU(imu13), dt, vf->settings.inputGainDiag(), vf->settings.stateGainDiag(vf->filterState.X.id.size()));
}
void orc_eqf_integrate_riccati_discrete(void* f, const double* imu13, double dt) {
    VIOFilter* vf = (VIOFilter*)f;
    vf->filterState.integrateRiccatiStateDiscrete(unpackIMU(imu13), dt, vf->settings.inputGainDiag(), vf->settings.stateGainDiag(vf->filterState.X.id.size()));
}
void orc_eqf_integrate_observer(void* f, const double* imu13, double dt, int discreteLift) {
    ((VIOFilter*)f)->filterState.integrateObserverState(unpackIMU(imu13), dt, discreteLift != 0);
}
void orc_eqf_vision_update(void* f, double stamp, const eqvio_camera* cam, const int* ids, const double* y, int M) {
    VIOFilter* vf = (VIOFilter*)f;
    vf->filterState.performVisionUpdate(makeMeasurement(stamp, cam, ids, y, M), vf->settings.outputGainVar(), vf->settings.useEquivariantOutput, vf->settings.useDiscreteInnovationLift);
}
void orc_eqf_remove_landmark_by_index(void* f, int idx) { ((VIOFilter*)f)->filterState.removeLandmarkByIndex(idx); }
void orc_eqf_add_landmarks(void* f, const int* ids, const double* p, int k, double var) {
    const double s0[23] = {0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0};
    ((VIOFilter*)f)->filterState.addNewLandmarks(unpackState(s0, ids, p, k).cameraLandmarks, var);
}
double orc_eqf_compute_nees(void* f, const double* sensor, const int* ids, const double* p, int N) {
    return ((VIOFilter*)f)->filterState.computeNEES(unpackState(sensor, ids, p, N));
}
// outlier statistics as removeOutliers computes them (src/VIOFilter.cpp:304-334): per state landmark i that is
// measured, abs[i] = ||y - yHat||, prob[i] = yTilde^T (C0 Sigma_ii C0^T)^-1 yTilde. Unmeasured -> -1.
void orc_filter_outlier_stats(void* f, const eqvio_camera* cam, const int* ids, const double* y, int M, double* absErr, double* probErr) {
    VIOFilter* vf = (VIOFilter*)f;
    const VisionMeasurement meas = makeMeasurement(0, cam, ids, y, M);
    const VIOState xiHat = vf->stateEstimate();
    const VisionMeasurement yHat = measureSystemState(xiHat, meas.cameraPtr);
    for (size_t i = 0; i < xiHat.cameraLandmarks.size(); ++i) {
        const int id = xiHat.cameraLandmarks[i].id;
        absErr[i] = probErr[i] = -1;
        if (!meas.camCoordinates.count(id))
            continue;
        const Vec2 yt = meas.camCoordinates.at(id) - yHat.camCoordinates.at(id);
        absErr[i] = yt.norm();
        const orc::M<2, 2> cov = vf->filterState.getOutputCovById(id, meas.camCoordinates.at(id), meas.cameraPtr);
        probErr[i] = dot(yt, inverse2(cov) * yt);
    }
}

// getOutputCovById (VIO_eqf.cpp:196-211) for every landmark of the state, state order, row-major 2 x 2 each: what eqf_output_cov_all returns
void orc_filter_output_cov_all(void* f, const eqvio_camera* cam, double* out4N) {
    VIOFilter* vf = (VIOFilter*)f;
    const int id0 = 0;
    const double y0[2] = {0.0, 0.0};
    const VisionMeasurement meas = makeMeasurement(0, cam, &id0, y0, 1); // for its camera pointer only: the pixel does not enter
    const VIO_eqf& e = vf->filterState;
    for (size_t i = 0; i < e.X.id.size(); ++i) {
        const orc::M<2, 2> cov = e.getOutputCovById(e.X.id[i], Vec2{}, meas.cameraPtr);
        out4N[4 * i + 0] = cov(0, 0), out4N[4 * i + 1] = cov(0, 1), out4N[4 * i + 2] = cov(1, 0), out4N[4 * i + 3] = cov(1, 1);
    }
}

// ---- EqF matrices (dense, column-major) for kernel-level parity
int orc_state_matrix_A(void* f, const double* imu13, double* out, int cap) {
    const VIO_eqf& e = ((VIOFilter*)f)->filterState;
    return copyMat(e.coordinateSuite->stateMatrixA(e.X, e.xi0, unpackIMU(imu13)), out, cap);
}
int orc_input_matrix_B(void* f, double* out, int cap) {
    const VIO_eqf& e = ((VIOFilter*)f)->filterState;
    return copyMat(e.coordinateSuite->inputMatrixB(e.X, e.xi0), out, cap);
}
int orc_output_matrix_C(void* f, const eqvio_camera* cam, const int* ids, const double* y, int M, int useEquivariance, double* out, int cap) {
    const VIO_eqf& e = ((VIOFilter*)f)->filterState;
    return copyMat(e.coordinateSuite->outputMatrixC(e.xi0, e.X, makeMeasurement(0, cam, ids, y, M), useEquivariance != 0), out, cap);
}
int orc_state_matrix_A_discrete(void* f, const double* imu13, double dt, double* out, int cap) {
    const VIO_eqf& e = ((VIOFilter*)f)->filterState;
    return copyMat(e.coordinateSuite->stateMatrixADiscrete(e.X, e.xi0, unpackIMU(imu13), dt), out, cap);
}
// state chart about xi0 of a given state (for error metrics in tests): eps = chart(xi, xi0)
int orc_state_chart(void* f, const double* sensor, const int* ids, const double* p, int N, double* out, int cap) {
    const VIO_eqf& e = ((VIOFilter*)f)->filterState;
    const DVec eps = e.coordinateSuite->stateChart(unpackState(sensor, ids, p, N), e.xi0);
    if ((int)eps.size() > cap)
        return -(int)eps.size();
    std::memcpy(out, eps.data(), eps.size() * sizeof(double));
    return (int)eps.size();
}
// integrateSystemFunction (src/mathematical/VIOState.cpp:28-68) on a packed state, in place
void orc_integrate_system_function(double* sensor, const int* ids, double* p, int N, const double* imu13, double dt) {
    const VIOState r = integrateSystemFunction(unpackState(sensor, ids, p, N), unpackIMU(imu13), dt);
    packState(r, sensor, nullptr, p, N);
}
// ||log(P1^-1 P2)|| for SE3 given as (qw,qx,qy,qz,x,y,z) — pose parity metric of SURVEY.md §8(d)
double orc_se3_log_dist(const double* a, const double* b) {
    const SE3 A(quat(a), vec3(a[4], a[5], a[6])), B(quat(b), vec3(b[4], b[5], b[6]));
    return SE3::log(A.inverse() * B).norm();
}

// ---- CPU baseline timing: run `reps` full frames (propagate + k observer steps + update) from the same
// starting state (teacher forced), return seconds per frame. mode: UpdateArithmetic.
double orc_bench_frame(void* f, const double* imu13_k, const double* dts, int k, double stamp, const eqvio_camera* cam, const int* ids, const double* y, int M, int mode, int reps) {
    VIOFilter* vf = (VIOFilter*)f;
    const VIO_eqf saved = vf->filterState;
    double total = 0;
    for (int r = 0; r < reps; ++r) {
        vf->filterState = saved;
        vf->filterState.arithmetic = (UpdateArithmetic)mode;
        const auto t0 = std::chrono::steady_clock::now();
        IMUVelocity acc = IMUVelocity::Zero();
        double T = 0;
        for (int i = 0; i < k; ++i) {
            acc = acc + unpackIMU(imu13_k + 13 * i) * dts[i];
            T += dts[i];
        }
        acc = acc * (1.0 / T);
        vf->filterState.integrateRiccatiStateFast(acc, T, vf->settings.inputGainDiag(), vf->settings.stateGainDiag(vf->filterState.X.id.size()));
        for (int i = 0; i < k; ++i)
            vf->filterState.integrateObserverState(unpackIMU(imu13_k + 13 * i), dts[i], vf->settings.useDiscreteVelocityLift);
        vf->filterState.performVisionUpdate(makeMeasurement(stamp, cam, ids, y, M), vf->settings.outputGainVar(), vf->settings.useEquivariantOutput, vf->settings.useDiscreteInnovationLift);
        total += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    vf->filterState = saved;
    return total / reps;
}

// camera model functions (tests/test_cameras.py)
void orc_cam_project(const eqvio_camera* cam, const double* p3, double* y2) {
    const Vec2 y = makeCamera(cam)->projectPoint(vec3(p3[0], p3[1], p3[2]));
    y2[0] = y(0);
    y2[1] = y(1);
}
void orc_cam_undistort(const eqvio_camera* cam, const double* y2, double* b3) {
    const Vec3 b = makeCamera(cam)->undistortPoint(vec2(y2[0], y2[1]));
    b3[0] = b(0);
    b3[1] = b(1);
    b3[2] = b(2);
}
void orc_cam_jacobian(const eqvio_camera* cam, const double* p3, double* J6_rowmajor) {
    const M<2, 3> J = makeCamera(cam)->projectionJacobian(vec3(p3[0], p3[1], p3[2]));
    for (int r = 0; r < 2; ++r)
        for (int c = 0; c < 3; ++c)
            J6_rowmajor[3 * r + c] = J(r, c);
}

// getFeaturePredictions (VIOFilter.cpp:247-252)
int orc_filter_get_feature_predictions(void* f, const eqvio_camera* cam, double stamp, int* ids, double* y, int cap) {
    const VisionMeasurement m = static_cast<VIOFilter*>(f)->getFeaturePredictions(makeCamera(cam), stamp);
    if ((int)m.camCoordinates.size() > cap)
        return -1;
    int k = 0;
    for (const auto& kv : m.camCoordinates) {
        ids[k] = kv.first;
        y[2 * k] = kv.second(0);
        y[2 * k + 1] = kv.second(1);
        ++k;
    }
    return k;
}

} // extern "C"
