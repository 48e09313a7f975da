// src/mathematical/VIO_eqf_mi355x.cpp — the reference-side binding of INTEGRATION.md §A, complete.
//
// It REPLACES the member bodies of src/mathematical/VIO_eqf.cpp:47-245 in the reference tree: every member of `struct VIO_eqf`
// (include/eqvio/mathematical/VIO_eqf.h:34-134) forwards to the C-ABI of include/eqf_hip.h; the struct keeps its data members
// (xi0, X, Sigma) as the HOST MIRROR of the device state. Written against the common subset of Eigen 3.4 / LiePP / GIFT and of the
// stand-in headers in tests/integration/standin/, so that it can be compiled, linked against libeqf_hip.so and run here, where
// those libraries are absent (tests/test_integration_stub.py). No file of the reference is needed to build it.
//
// Coherence rules (the whole protocol):
//  * every member call first makes the device current (`ensure`): creates the context on first use, uploads the host members
//    when they were assigned directly (`markHostEdited()`, or a fresh / copied filter), grows the capacity when needed;
//  * every mutating member leaves the DEVICE ahead (`twin.deviceNewer`); ids and container sizes of xi0 / X are kept current
//    on the host eagerly (VIOFilter.cpp reads `filterState.X.id` directly), numbers lazily;
//  * `pull()` brings the numbers back: VIOFilter::viewEqFState() calls it (the writers' only access path, VIOWriter.cpp:162-222);
//  * copying a VIO_eqf clones the device twin (DeviceTwin's copy constructor), moving it moves the context.
#include "eqvio/mathematical/VIO_eqf.h"

#include <algorithm>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <utility>

#include "eqf_hip.h"

namespace {
constexpr int kSensorDim = VIOSensorState::CompDim; // 21

void check(int rc, const char* what) {
    // EQF_E_NONFINITE is the reference's assert(!Sigma.hasNaN()); the others have no reference analogue and are errors
    if (rc < 0)
        throw std::runtime_error(std::string(what) + ": " + eqf_error_string(rc));
}
int chartOf(const EqFCoordinateSuite* suite) {
    if (suite == &EqFCoordinateSuite_euclid)
        return EQVIO_COORD_EUCLIDEAN;
    if (suite == &EqFCoordinateSuite_invdepth)
        return EQVIO_COORD_INVDEPTH;
    if (suite == &EqFCoordinateSuite_normal)
        return EQVIO_COORD_NORMAL;
    throw std::invalid_argument("VIO_eqf: unknown coordinate suite");
}
int deviceIndex() {
    const char* e = std::getenv("EQVIO_MI355X_DEVICE");
    return e ? std::atoi(e) : 0;
}

// eqvio_types.h: a pose is (qw, qx, qy, qz, x, y, z); a sensor state / sensor group element is 23 doubles
void packSE3(const liepp::SE3d& P, double* q7) {
    const Eigen::Quaterniond q = P.R.asQuaternion();
    q7[0] = q.w(), q7[1] = q.x(), q7[2] = q.y(), q7[3] = q.z();
    std::copy(P.x.data(), P.x.data() + 3, q7 + 4);
}
liepp::SE3d unpackSE3(const double* q7) {
    return liepp::SE3d(liepp::SO3d(Eigen::Quaterniond(q7[0], q7[1], q7[2], q7[3])), Eigen::Vector3d(q7[4], q7[5], q7[6]));
}
void packSensor(const VIOSensorState& s, double* v23) {
    std::copy(s.inputBias.data(), s.inputBias.data() + 6, v23);
    packSE3(s.pose, v23 + 6);
    std::copy(s.velocity.data(), s.velocity.data() + 3, v23 + 13);
    packSE3(s.cameraOffset, v23 + 16);
}
void unpackSensor(const double* v23, VIOSensorState& s) {
    std::copy(v23, v23 + 6, s.inputBias.data());
    s.pose = unpackSE3(v23 + 6);
    std::copy(v23 + 13, v23 + 16, s.velocity.data());
    s.cameraOffset = unpackSE3(v23 + 16);
}
void packGroupSensor(const VIOGroup& X, double* v23) {
    std::copy(X.beta.data(), X.beta.data() + 6, v23);
    packSE3(X.A, v23 + 6);
    std::copy(X.w.data(), X.w.data() + 3, v23 + 13);
    packSE3(X.B, v23 + 16);
}
void packImu(const IMUVelocity& v, double* imu13) { // (stamp, gyr, acc, gyrBiasVel, accBiasVel)
    imu13[0] = v.stamp;
    std::copy(v.gyr.data(), v.gyr.data() + 3, imu13 + 1);
    std::copy(v.acc.data(), v.acc.data() + 3, imu13 + 4);
    std::copy(v.gyrBiasVel.data(), v.gyrBiasVel.data() + 3, imu13 + 7);
    std::copy(v.accBiasVel.data(), v.accBiasVel.data() + 3, imu13 + 10);
}
// constructInputGainMatrix / constructStateGainMatrix (VIOFilterSettings.h:160-201) are diagonal, and constant within each 3-block
// of the state gain: the ABI takes the 12 + 8 numbers (eqf_hip.h, eqf_integrate_riccati_fast)
void packGains(const Eigen::Matrix<double, 12, 12>& Q, const Eigen::MatrixXd& P, double* Qd12, double* Pd8) {
    for (int i = 0; i < 12; ++i)
        Qd12[i] = Q(i, i);
    for (int b = 0; b < 7; ++b)
        Pd8[b] = P(3 * b, 3 * b);
    Pd8[7] = P.rows() > kSensorDim ? P(kSensorDim, kSensorDim) : 0.0;
}
eqvio_camera toEqvioCamera(const GIFT::GICamera& cam) {
    eqvio_camera c{};
    c.width = cam.imageSize.width, c.height = cam.imageSize.height;
    c.fx = cam.fx, c.fy = cam.fy, c.cx = cam.cx, c.cy = cam.cy;
    c.model = EQVIO_CAMERA_PINHOLE;
    if (const auto* radtan = dynamic_cast<const GIFT::StandardCamera*>(&cam)) {
        c.model = EQVIO_CAMERA_RADTAN;
        for (size_t i = 0; i < radtan->dist.size() && i < 5; ++i)
            c.dist[i] = radtan->dist[i];
    } else if (const auto* equi = dynamic_cast<const GIFT::EquidistantCamera*>(&cam)) {
        c.model = EQVIO_CAMERA_EQUIDISTANT;
        std::copy(equi->dist.begin(), equi->dist.end(), c.dist);
    }
    return c;
}

void upload(const VIO_eqf& f, eqf_ctx* ctx) { // host members -> device
    const int N = (int)f.X.id.size();
    if ((int)f.xi0.cameraLandmarks.size() != N || (int)f.X.Q.size() != N || f.Sigma.rows() != kSensorDim + 3 * N || f.Sigma.cols() != f.Sigma.rows())
        throw std::invalid_argument("VIO_eqf: xi0 / X / Sigma sizes disagree");
    double xi0s[23], Xs[23];
    packSensor(f.xi0.sensor, xi0s);
    packGroupSensor(f.X, Xs);
    std::vector<double> q0(3 * (size_t)N), Q(5 * (size_t)N);
    for (int i = 0; i < N; ++i) {
        if (f.xi0.cameraLandmarks[i].id != f.X.id[i])
            throw std::invalid_argument("VIO_eqf: xi0.cameraLandmarks and X.id are ordered differently");
        std::copy(f.xi0.cameraLandmarks[i].p.data(), f.xi0.cameraLandmarks[i].p.data() + 3, &q0[3 * i]);
        const Eigen::Quaterniond q = f.X.Q[i].R.asQuaternion();
        Q[5 * i] = q.w(), Q[5 * i + 1] = q.x(), Q[5 * i + 2] = q.y(), Q[5 * i + 3] = q.z(), Q[5 * i + 4] = f.X.Q[i].a;
    }
    check(eqf_set_state(ctx, xi0s, Xs, f.X.id.data(), q0.data(), Q.data(), N), "eqf_set_state");
    check(eqf_set_sigma(ctx, f.Sigma.data(), f.Sigma.rows()), "eqf_set_sigma"); // Eigen::MatrixXd is column-major, as the ABI
}
void download(eqf_ctx* ctx, VIO_eqf& f) { // device -> host members
    const int N = eqf_num_landmarks(ctx);
    double xi0s[23], Xs[23];
    std::vector<int> ids(N);
    std::vector<double> q0(3 * (size_t)N), Q(5 * (size_t)N);
    check(eqf_get_state(ctx, xi0s, Xs, ids.data(), q0.data(), Q.data(), N), "eqf_get_state");
    unpackSensor(xi0s, f.xi0.sensor);
    std::copy(Xs, Xs + 6, f.X.beta.data());
    f.X.A = unpackSE3(Xs + 6);
    std::copy(Xs + 13, Xs + 16, f.X.w.data());
    f.X.B = unpackSE3(Xs + 16);
    f.xi0.cameraLandmarks.resize(N);
    f.X.Q.resize(N);
    f.X.id = ids;
    for (int i = 0; i < N; ++i) {
        f.xi0.cameraLandmarks[i].id = ids[i];
        std::copy(&q0[3 * i], &q0[3 * i] + 3, f.xi0.cameraLandmarks[i].p.data());
        f.X.Q[i].R = liepp::SO3d(Eigen::Quaterniond(Q[5 * i], Q[5 * i + 1], Q[5 * i + 2], Q[5 * i + 3]));
        f.X.Q[i].a = Q[5 * i + 4];
    }
    const int n = kSensorDim + 3 * N;
    f.Sigma.resize(n, n);
    check(eqf_get_sigma(ctx, f.Sigma.data(), n), "eqf_get_sigma");
}

// the recorded integrateRiccatiStateFast + k x integrateObserverState of one integrateUpToTime as ONE device call (DeviceTwin: "Deferred propagation")
void flushPending(eqvio_mi355x::DeviceTwin& t) {
    if (!t.pendingRiccati)
        return;
    t.pendingRiccati = false;
    const int k = (int)t.pendingDts.size();
    check(eqf_propagate_fast(t.ctx, t.pendingMean, t.pendingDt, t.pendingQd, t.pendingPd8, t.pendingImu.data(), t.pendingDts.data(), k, t.pendingDiscrete ? 1 : 0), "deferred propagation");
    t.pendingImu.clear(), t.pendingDts.clear();
}
// make the device current for `f`; `extra` landmarks are about to be appended. `keepPending`: the caller is an observer step that joins the deferred propagation
eqf_ctx* ensure(const VIO_eqf& f, int extra = 0, bool keepPending = false) {
    eqvio_mi355x::DeviceTwin& t = f.twin;
    if (!keepPending && t.ctx)
        flushPending(t); // (before anything else looks at the device, or rebuilds / re-uploads it)
    const int chart = chartOf(f.coordinateSuite);
    const int need = (int)f.X.id.size() + extra;
    if (t.ctx && (t.chart != chart || t.capacity < need)) { // another chart or more landmarks than the context holds: rebuild it
        if (t.deviceNewer && !t.hostEdited)
            f.pull();
        eqf_destroy(t.ctx);
        t.ctx = nullptr;
        t.hostEdited = true;
    }
    if (!t.ctx) {
        t.capacity = std::max(64, 2 * need);
        t.chart = chart;
        check(eqf_create(&t.ctx, deviceIndex(), t.capacity, chart), "eqf_create");
        t.hostEdited = true;
    }
    if (t.hostEdited) {
        upload(f, t.ctx);
        t.hostEdited = false;
        t.deviceNewer = false;
        t.outCovValid = false;
    }
    return t.ctx;
}
} // namespace

// ---------------------------------------------------------------- the device twin
namespace eqvio_mi355x {
DeviceTwin::DeviceTwin(const DeviceTwin& o) : chart(o.chart), capacity(o.capacity), deviceNewer(o.deviceNewer), hostEdited(o.hostEdited) {
    if (o.ctx)
        flushPending(const_cast<DeviceTwin&>(o)); // a recorded propagation belongs to the state that is being copied
    if (o.ctx && o.deviceNewer && !o.hostEdited) { // the source's host members are behind its device: clone device -> device (through the host)
        VIO_eqf scratch;
        download(o.ctx, scratch);
        check(eqf_create(&ctx, deviceIndex(), capacity, chart), "eqf_create");
        upload(scratch, ctx);
    } else { // the host members that were copied alongside are current: upload them on first use
        hostEdited = true;
        deviceNewer = false;
    }
}
DeviceTwin& DeviceTwin::operator=(const DeviceTwin& o) {
    if (this != &o) {
        DeviceTwin tmp(o);
        *this = std::move(tmp);
    }
    return *this;
}
static void movePending(DeviceTwin& to, DeviceTwin& from) {
    to.pendingRiccati = from.pendingRiccati, to.pendingDiscrete = from.pendingDiscrete, to.pendingDt = from.pendingDt;
    std::copy(from.pendingMean, from.pendingMean + 13, to.pendingMean);
    std::copy(from.pendingQd, from.pendingQd + 12, to.pendingQd);
    std::copy(from.pendingPd8, from.pendingPd8 + 8, to.pendingPd8);
    to.pendingImu.swap(from.pendingImu), to.pendingDts.swap(from.pendingDts);
    from.pendingRiccati = false;
}
DeviceTwin::DeviceTwin(DeviceTwin&& o) noexcept : ctx(o.ctx), chart(o.chart), capacity(o.capacity), deviceNewer(o.deviceNewer), hostEdited(o.hostEdited) {
    movePending(*this, o);
    o.ctx = nullptr;
}
DeviceTwin& DeviceTwin::operator=(DeviceTwin&& o) noexcept {
    if (this != &o) {
        if (ctx)
            eqf_destroy(ctx);
        ctx = o.ctx, chart = o.chart, capacity = o.capacity, deviceNewer = o.deviceNewer, hostEdited = o.hostEdited;
        movePending(*this, o);
        o.ctx = nullptr;
    }
    return *this;
}
DeviceTwin::~DeviceTwin() {
    if (ctx)
        eqf_destroy(ctx);
}
} // namespace eqvio_mi355x

void VIO_eqf::pull() const {
    if (twin.ctx)
        flushPending(twin);
    if (twin.ctx && twin.deviceNewer && !twin.hostEdited) {
        download(twin.ctx, const_cast<VIO_eqf&>(*this)); // the host members are a cache of the device state
        twin.deviceNewer = false;
    }
}

// ---------------------------------------------------------------- propagation (VIO_eqf.cpp:47-103)
void VIO_eqf::integrateObserverState(const IMUVelocity& imuVelocity, const double& dt, const bool& discreteLift) {
    double imu[13];
    packImu(imuVelocity, imu);
    if (twin.ctx && twin.pendingRiccati && (twin.pendingDts.empty() || twin.pendingDiscrete == discreteLift)) { // joins the propagation recorded in front of it
        ensure(*this, 0, true);
        twin.pendingDiscrete = discreteLift;
        twin.pendingImu.insert(twin.pendingImu.end(), imu, imu + 13);
        twin.pendingDts.push_back(dt);
        twin.touch();
        return;
    }
    eqf_ctx* ctx = ensure(*this);
    check(eqf_integrate_observer(ctx, imu, &dt, 1, discreteLift ? 1 : 0), "integrateObserverState");
    twin.touch();
}
void VIO_eqf::integrateRiccatiStateFast(const IMUVelocity& imuVelocity, const double& dt, const Eigen::Matrix<double, 12, 12>& inputGainMatrix,
                                        const Eigen::MatrixXd& stateGainMatrix) {
    ensure(*this); // (flushes whatever was recorded before)
    // recorded, not issued: the observer steps that VIOFilter::integrateUpToTime makes next join it, and the first other member call issues everything as one launch
    packImu(imuVelocity, twin.pendingMean);
    packGains(inputGainMatrix, stateGainMatrix, twin.pendingQd, twin.pendingPd8);
    twin.pendingDt = dt;
    twin.pendingRiccati = true;
    twin.pendingImu.clear(), twin.pendingDts.clear();
    twin.touch();
}
void VIO_eqf::integrateRiccatiStateAccurate(const IMUVelocity& imuVelocity, const double& dt, const Eigen::Matrix<double, 12, 12>& inputGainMatrix,
                                            const Eigen::MatrixXd& stateGainMatrix) {
    eqf_ctx* ctx = ensure(*this);
    double imu[13], Qd[12], Pd8[8];
    packImu(imuVelocity, imu);
    packGains(inputGainMatrix, stateGainMatrix, Qd, Pd8);
    check(eqf_integrate_riccati_accurate(ctx, imu, dt, Qd, Pd8), "integrateRiccatiStateAccurate");
    twin.touch();
}
void VIO_eqf::integrateRiccatiStateDiscrete(const IMUVelocity& imuVelocity, const double& dt, const Eigen::Matrix<double, 12, 12>& inputGainMatrix,
                                            const Eigen::MatrixXd& stateGainMatrix) {
    eqf_ctx* ctx = ensure(*this);
    double imu[13], Qd[12], Pd8[8];
    packImu(imuVelocity, imu);
    packGains(inputGainMatrix, stateGainMatrix, Qd, Pd8);
    check(eqf_integrate_riccati_discrete(ctx, imu, dt, Qd, Pd8), "integrateRiccatiStateDiscrete");
    twin.touch();
}

// ---------------------------------------------------------------- vision update (VIO_eqf.cpp:105-135)
void VIO_eqf::performVisionUpdate(const VisionMeasurement& measurement, const Eigen::MatrixXd& outputGainMatrix, const bool& useEquivariantOutput,
                                  const bool& discreteCorrection) {
    if (measurement.camCoordinates.empty())
        return;
    eqf_ctx* ctx = ensure(*this);
    std::vector<int> ids;
    std::vector<double> px;
    for (const auto& [id, y] : measurement.camCoordinates) { // std::map iterates in ascending id: the reference's row order
        ids.push_back(id);
        px.push_back(y.x());
        px.push_back(y.y());
    }
    const eqvio_camera cam = toEqvioCamera(*measurement.cameraPtr);
    // constructOutputGainMatrix (VIOFilterSettings.h:203-206) = measurementNoise^2 * I
    check(eqf_vision_update(ctx, &cam, ids.data(), px.data(), (int)ids.size(), outputGainMatrix(0, 0), useEquivariantOutput ? 1 : 0, discreteCorrection ? 1 : 0),
          "performVisionUpdate");
    twin.touch();
}

// ---------------------------------------------------------------- read-outs (VIO_eqf.cpp:137-170, 188-211)
VIOState VIO_eqf::stateEstimate() const {
    eqf_ctx* ctx = ensure(*this);
    const int N = (int)X.id.size();
    double sensor[23];
    std::vector<int> ids(N);
    std::vector<double> p(3 * (size_t)N);
    check(eqf_state_estimate(ctx, sensor, ids.data(), p.data(), N), "stateEstimate");
    VIOState xi;
    unpackSensor(sensor, xi.sensor);
    xi.cameraLandmarks.resize(N);
    for (int i = 0; i < N; ++i) {
        xi.cameraLandmarks[i].id = ids[i];
        std::copy(&p[3 * i], &p[3 * i] + 3, xi.cameraLandmarks[i].p.data());
    }
    return xi;
}
double VIO_eqf::computeNEES(const VIOState& trueState) const {
    eqf_ctx* ctx = ensure(*this);
    double sensor[23];
    packSensor(trueState.sensor, sensor);
    std::vector<int> ids;
    std::vector<double> p;
    for (const Landmark& lm : trueState.cameraLandmarks) {
        ids.push_back(lm.id);
        p.insert(p.end(), lm.p.data(), lm.p.data() + 3);
    }
    double nees = 0.0;
    check(eqf_compute_nees(ctx, sensor, ids.data(), p.data(), (int)ids.size(), &nees), "computeNEES");
    return nees;
}
Eigen::Matrix3d VIO_eqf::getLandmarkCovById(const int& id) const {
    eqf_ctx* ctx = ensure(*this);
    const auto it = std::find(X.id.begin(), X.id.end(), id);
    if (it == X.id.end())
        throw std::out_of_range("getLandmarkCovById: unknown id");
    const int r = kSensorDim + 3 * (int)(it - X.id.begin());
    Eigen::Matrix3d cov;
    check(eqf_get_sigma_block(ctx, r, r, 3, 3, cov.data()), "getLandmarkCovById");
    return cov;
}
Eigen::Matrix2d VIO_eqf::getOutputCovById(const int& id, const Eigen::Vector2d&, const GIFT::GICameraPtr& camPtr) const {
    // C0i Sigma_ii C0i^T with C0i = outputMatrixCi (not the equivariant C*; the pixel does not enter, VIO_eqf.cpp:196-211). The reference's removeOutliers
    // asks for one landmark per call (VIOFilter.cpp:304-334): one device round trip per landmark made the unchanged VIOFilter.cpp run at 111 frames/s with
    // 200 landmarks. The first call after the state changed now fetches the covariances of ALL landmarks (eqf_output_cov_all: one kernel, one wait);
    // the following calls of the frame read the cache. (eqf_outlier_stats is the fused form this repo's own VIOFilter mirror uses instead.)
    eqf_ctx* ctx = ensure(*this);
    eqvio_mi355x::DeviceTwin& t = twin;
    // removeOutliers walks the landmarks in state order (VIOFilter.cpp:316-334): the id asked for is usually the one behind the last one
    size_t idx = t.outCovLast + 1;
    if (idx >= X.id.size() || X.id[idx] != id) {
        const auto it = std::find(X.id.begin(), X.id.end(), id);
        if (it == X.id.end())
            throw std::out_of_range("getOutputCovById: unknown id");
        idx = (size_t)(it - X.id.begin());
    }
    t.outCovLast = idx;
    const eqvio_camera cam = toEqvioCamera(*camPtr);
    const double key[10] = {(double)cam.model, cam.fx, cam.fy, cam.cx, cam.cy, cam.dist[0], cam.dist[1], cam.dist[2], cam.dist[3], cam.dist[4]};
    if (!t.outCovValid || t.outCov.size() != 4 * X.id.size() || !std::equal(key, key + 10, t.outCovCam)) {
        t.outCov.assign(4 * X.id.size(), 0.0);
        check(eqf_output_cov_all(ctx, &cam, t.outCov.data()), "getOutputCovById");
        std::copy(key, key + 10, t.outCovCam);
        t.outCovValid = true;
    }
    const double* v = &t.outCov[4 * idx];
    Eigen::Matrix2d out;
    out(0, 0) = v[0], out(0, 1) = v[1], out(1, 0) = v[2], out(1, 1) = v[3];
    return out;
}

// ---------------------------------------------------------------- landmark bookkeeping (VIO_eqf.cpp:172-186, 213-245)
void VIO_eqf::addNewLandmarks(std::vector<Landmark>& newLandmarks, const Eigen::MatrixXd& newLandmarkCov) {
    if (newLandmarks.empty())
        return;
    const int k = (int)newLandmarks.size();
    eqf_ctx* ctx = ensure(*this, k);
    std::vector<int> ids(k);
    std::vector<double> p(3 * (size_t)k);
    for (int i = 0; i < k; ++i) {
        ids[i] = newLandmarks[i].id;
        std::copy(newLandmarks[i].p.data(), newLandmarks[i].p.data() + 3, &p[3 * i]);
    }
    // the reference's callers pass initialPointVariance * I (VIOFilter.cpp:129-130, 274-276): the ABI takes the scalar
    for (int i = 0; i < 3 * k; ++i)
        for (int j = 0; j < 3 * k; ++j)
            if (newLandmarkCov(i, j) != (i == j ? newLandmarkCov(0, 0) : 0.0))
                throw std::invalid_argument("addNewLandmarks: newLandmarkCov must be a multiple of the identity");
    check(eqf_add_landmarks(ctx, ids.data(), p.data(), k, newLandmarkCov(0, 0)), "addNewLandmarks");
    xi0.cameraLandmarks.insert(xi0.cameraLandmarks.end(), newLandmarks.begin(), newLandmarks.end());
    X.id.insert(X.id.end(), ids.begin(), ids.end());
    X.Q.resize(X.id.size());
    twin.touch();
}
void VIO_eqf::removeLandmarkByIndex(const int& idx) {
    eqf_ctx* ctx = ensure(*this);
    check(eqf_remove_landmarks(ctx, &idx, 1), "removeLandmarkByIndex");
    xi0.cameraLandmarks.erase(xi0.cameraLandmarks.begin() + idx);
    X.id.erase(X.id.begin() + idx);
    X.Q.erase(X.Q.begin() + idx);
    twin.touch();
}
void VIO_eqf::removeLandmarkById(const int& id) {
    const auto it = std::find(X.id.begin(), X.id.end(), id);
    if (it == X.id.end())
        throw std::out_of_range("removeLandmarkById: unknown id");
    removeLandmarkByIndex((int)(it - X.id.begin()));
}
void VIO_eqf::removeInvalidLandmarks() {
    eqf_ctx* ctx = ensure(*this);
    const int removed = eqf_remove_invalid_landmarks(ctx);
    check(removed, "removeInvalidLandmarks");
    if (removed > 0) { // which ones: ask the device for the surviving ids and drop the others from the host containers
        std::vector<int> ids(X.id.size());
        const int N = eqf_get_ids(ctx, ids.data(), (int)ids.size());
        check(N, "eqf_get_ids");
        ids.resize(N);
        size_t keep = 0;
        for (size_t i = 0; i < X.id.size(); ++i)
            if (keep < ids.size() && X.id[i] == ids[keep]) {
                xi0.cameraLandmarks[keep] = xi0.cameraLandmarks[i];
                X.Q[keep] = X.Q[i];
                X.id[keep] = X.id[i];
                ++keep;
            }
        xi0.cameraLandmarks.resize(keep);
        X.Q.resize(keep);
        X.id.resize(keep);
    }
    twin.touch();
}

// ---------------------------------------------------------------- fused entry points (optional; VIOFilter_mi355x_hunks.hpp with mi355xFused)
namespace {
void flattenMeasurement(const VisionMeasurement& measurement, std::vector<int>& ids, std::vector<double>& px) {
    ids.clear(), px.clear();
    for (const auto& [id, y] : measurement.camCoordinates) { // ascending id: the reference's row order
        ids.push_back(id);
        px.push_back(y.x());
        px.push_back(y.y());
    }
}
} // namespace
void VIO_eqf::propagateFast(const IMUVelocity& meanVelocity, const double& dtTotal, const double (&Qd)[12], const double (&Pd8)[8], const std::vector<IMUVelocity>& samples,
                            const std::vector<double>& dts, const bool& discreteLift) {
    eqf_ctx* ctx = ensure(*this);
    double mean[13];
    packImu(meanVelocity, mean);
    std::vector<double> all(13 * samples.size());
    for (size_t i = 0; i < samples.size(); ++i)
        packImu(samples[i], &all[13 * i]);
    check(eqf_propagate_fast(ctx, mean, dtTotal, Qd, Pd8, all.data(), dts.data(), (int)samples.size(), discreteLift ? 1 : 0), "propagateFast");
    twin.touch();
}
void VIO_eqf::stageMeasurement(const VisionMeasurement& measurement) {
    if (measurement.camCoordinates.empty() || X.id.empty())
        return;
    eqf_ctx* ctx = ensure(*this);
    std::vector<int> ids;
    std::vector<double> px;
    flattenMeasurement(measurement, ids, px);
    check(eqf_stage_measurement(ctx, ids.data(), px.data(), (int)ids.size()), "stageMeasurement");
}
int VIO_eqf::statsThenUpdate(const VisionMeasurement& measurement, const double& thrAbs, const double& thrProb, const long& maxOutliers, const double& outputGainVariance,
                             const bool& useEquivariantOutput, const bool& discreteCorrection, std::vector<double>& absErr, std::vector<double>& probErr) {
    eqf_ctx* ctx = ensure(*this);
    const int N = (int)X.id.size();
    absErr.assign(N, -1.0), probErr.assign(N, -1.0);
    std::vector<double> depth2(N, 0.0);
    std::vector<int> ids;
    std::vector<double> px;
    flattenMeasurement(measurement, ids, px);
    const eqvio_camera cam = toEqvioCamera(*measurement.cameraPtr);
    int updated = 0;
    if (maxOutliers < 0) {
        check(eqf_stats_then_update(ctx, &cam, ids.data(), px.data(), (int)ids.size(), thrAbs, thrProb, outputGainVariance, useEquivariantOutput ? 1 : 0, discreteCorrection ? 1 : 0,
                                    absErr.data(), probErr.data(), depth2.data(), &updated),
              "statsThenUpdate");
    } else {
        std::vector<int> removed(N + 1);
        int nRemoved = 0;
        check(eqf_stats_select_update(ctx, &cam, ids.data(), px.data(), (int)ids.size(), thrAbs, thrProb, (int)std::min<long>(maxOutliers, 1 << 30), outputGainVariance,
                                      useEquivariantOutput ? 1 : 0, discreteCorrection ? 1 : 0, absErr.data(), probErr.data(), depth2.data(), &updated, removed.data(), &nRemoved),
              "statsSelectUpdate");
        for (int t = nRemoved - 1; t >= 0; --t) { // the device removed these landmarks (ascending indices): follow on the host containers
            xi0.cameraLandmarks.erase(xi0.cameraLandmarks.begin() + removed[t]);
            X.id.erase(X.id.begin() + removed[t]);
            X.Q.erase(X.Q.begin() + removed[t]);
        }
    }
    if (updated == 1)
        twin.touch();
    return updated;
}
