// Host-side control logic of the filter (mirror of src/VIOFilter.cpp and the VIO_eqf bookkeeping of
// src/mathematical/VIO_eqf.cpp); every matrix operation goes through the C-ABI of include/eqf_hip.h.
#include "VIOFilter.hpp"
#include "../csrc/host_prof.hpp"
#include <optional>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>

namespace eqvio_amd {
namespace { constexpr double GRAVITY_CONSTANT_ = 9.80665; } // include/eqvio/mathematical/IMUVelocity.h:26

using namespace eqf;

thread_local LoopTimer loopTimer;

// ---------------------------------------------------------------- LoopTimer (src/LoopTimer.cpp:20-43)
void LoopTimer::startTiming(const std::string& label) { timerStartPoints.at(label) = timer_clock::now(); }
void LoopTimer::endTiming(const std::string& label) {
    const timer_clock::time_point now = timer_clock::now();
    currentLoopTimingData.timings[label] = now - timerStartPoints.at(label);
}
void LoopTimer::startLoop() {
    for (const auto& kv : timerStartPoints)
        currentLoopTimingData.timings[kv.first] = timer_duration(0);
    currentLoopTimingData.loopTimeStart = timer_clock::now() - timerOrigin;
}
void LoopTimer::initialise(const std::vector<std::string>& headers) {
    const timer_clock::time_point now = timer_clock::now();
    for (const std::string& h : headers)
        timerStartPoints[h] = now;
}

// ---------------------------------------------------------------- small value types
std::vector<int> VIOState::getIds() const {
    std::vector<int> ids(cameraLandmarks.size());
    std::transform(cameraLandmarks.begin(), cameraLandmarks.end(), ids.begin(), [](const Landmark& lm) { return lm.id; });
    return ids;
}
std::vector<int> VisionMeasurement::getIds() const { return flatIds(); }
void VisionMeasurement::refreshFlat() const {
    // camCoordinates is a public std::map (the reference's shape): a caller may change pixel values or swap an interior id without this object
    // noticing. The cache is therefore trusted only after ONE walk over the map that compares every id and both pixel values (O(M), no allocation;
    // about as long as the two allocations a rebuild would cost); the first difference rebuilds it.
    const size_t n = camCoordinates.size();
    if (flatTrusted_.on && flatN_ == n && flatIds_.size() == n)
        return; // validated at the start of the call that holds this measurement (Validated)
    if (flatN_ == n && flatIds_.size() == n) {
        size_t j = 0;
        for (auto it = camCoordinates.begin(); it != camCoordinates.end(); ++it, ++j)
            if (flatIds_[j] != it->first || flatY_[2 * j] != it->second[0] || flatY_[2 * j + 1] != it->second[1])
                break;
        if (j == n)
            return;
    }
    ++flatRebuilds_;
    flatIds_.resize(n);
    flatY_.resize(2 * n);
    size_t j = 0;
    for (const auto& kv : camCoordinates) {
        flatIds_[j] = kv.first;
        flatY_[2 * j] = kv.second[0];
        flatY_[2 * j + 1] = kv.second[1];
        ++j;
    }
    flatN_ = n;
}
IMUVelocity IMUVelocity::operator+(const IMUVelocity& o) const { // src/mathematical/IMUVelocity.cpp:42-50
    IMUVelocity r;
    r.stamp = (stamp > 0) ? stamp : o.stamp;
    r.gyr = gyr + o.gyr;
    r.acc = acc + o.acc;
    r.gyrBiasVel = gyrBiasVel + o.gyrBiasVel;
    r.accBiasVel = accBiasVel + o.accBiasVel;
    return r;
}
IMUVelocity IMUVelocity::operator*(const double& c) const { // IMUVelocity.cpp:69-77
    IMUVelocity r;
    r.stamp = stamp;
    r.gyr = c * gyr;
    r.acc = c * acc;
    r.gyrBiasVel = c * gyrBiasVel;
    r.accBiasVel = c * accBiasVel;
    return r;
}
void IMUVelocity::pack(double* v) const {
    v[0] = stamp;
    const V3 parts[4] = {gyr, acc, gyrBiasVel, accBiasVel};
    for (int b = 0; b < 4; ++b) {
        v[1 + 3 * b] = parts[b].x;
        v[2 + 3 * b] = parts[b].y;
        v[3 + 3 * b] = parts[b].z;
    }
}

namespace {
void packPose(const Pose& p, double* q7) {
    q7[0] = p.R.w;
    q7[1] = p.R.x;
    q7[2] = p.R.y;
    q7[3] = p.R.z;
    q7[4] = p.x.x;
    q7[5] = p.x.y;
    q7[6] = p.x.z;
}
Pose unpackPose(const double* q7) { return Pose{Qt{q7[0], q7[1], q7[2], q7[3]}, V3{q7[4], q7[5], q7[6]}}; }
void packSensor(const VIOSensorState& s, double* d) {
    std::memcpy(d, s.inputBias.data(), sizeof(double) * 6);
    packPose(s.pose, d + 6);
    d[13] = s.velocity.x;
    d[14] = s.velocity.y;
    d[15] = s.velocity.z;
    packPose(s.cameraOffset, d + 16);
}
VIOSensorState unpackSensor(const double* d) {
    VIOSensorState s;
    std::memcpy(s.inputBias.data(), d, sizeof(double) * 6);
    s.pose = unpackPose(d + 6);
    s.velocity = V3{d[13], d[14], d[15]};
    s.cameraOffset = unpackPose(d + 16);
    return s;
}
// the measurement as the two flat arrays the C-ABI takes: one validated view per call (VisionMeasurement::flat), no copies
struct FlatMeas {
    const std::vector<int>& ids;
    const std::vector<double>& y;
    explicit FlatMeas(const VisionMeasurement& m) : FlatMeas(m.flat()) {}
    FlatMeas(std::pair<const std::vector<int>*, const std::vector<double>*> v) : ids(*v.first), y(*v.second) {}
};
} // namespace

// ---------------------------------------------------------------- VIO_eqf (device backed)
VIO_eqf::~VIO_eqf() {
    if (ctx)
        eqf_destroy(ctx);
}
void VIO_eqf::check(int rc, const char* what) const {
    if (rc != 0)
        throw std::runtime_error(std::string("eqf_hip: ") + what + ": " + eqf_error_string(rc));
}
void VIO_eqf::create(int device, int maxLandmarks, CoordinateChoice cc) {
    if (ctx) {
        eqf_destroy(ctx);
        ctx = nullptr;
    }
    coordinateChoice = cc;
    check(eqf_create(&ctx, device, maxLandmarks, (int)cc), "eqf_create");
    ids_.clear();
}
void VIO_eqf::set(const VIOState& xi0, const VIOGroup& X) {
    settleInvalid();
    const int N = (int)xi0.cameraLandmarks.size();
    if ((int)X.Q.size() != N || (int)X.id.size() != N)
        throw std::invalid_argument("VIO_eqf::set: xi0 and X landmark counts differ");
    double s[23], g[23];
    packSensor(xi0.sensor, s);
    std::memcpy(g, X.beta.data(), sizeof(double) * 6);
    packPose(X.A, g + 6);
    g[13] = X.w.x;
    g[14] = X.w.y;
    g[15] = X.w.z;
    packPose(X.B, g + 16);
    std::vector<double> q0(3 * N), Q(5 * N);
    ids_.resize(N);
    for (int i = 0; i < N; ++i) {
        if (X.id[i] != xi0.cameraLandmarks[i].id)
            throw std::invalid_argument("VIO_eqf::set: xi0 and X ids are not aligned");
        ids_[i] = X.id[i];
        q0[3 * i] = xi0.cameraLandmarks[i].p.x;
        q0[3 * i + 1] = xi0.cameraLandmarks[i].p.y;
        q0[3 * i + 2] = xi0.cameraLandmarks[i].p.z;
        Q[5 * i] = X.Q[i].R.w;
        Q[5 * i + 1] = X.Q[i].R.x;
        Q[5 * i + 2] = X.Q[i].R.y;
        Q[5 * i + 3] = X.Q[i].R.z;
        Q[5 * i + 4] = X.Q[i].a;
    }
    check(eqf_set_state(ctx, s, g, ids_.data(), q0.data(), Q.data(), N), "eqf_set_state");
}
VIOState VIO_eqf::xi0() const {
    const int N = numLandmarks();
    double s[23], g[23];
    std::vector<int> ids(N + 1);
    std::vector<double> q0(3 * N + 3), Q(5 * N + 5);
    const int rc = eqf_get_state(ctx, s, g, ids.data(), q0.data(), Q.data(), N);
    if (rc < 0)
        check(rc, "eqf_get_state");
    VIOState xi;
    xi.sensor = unpackSensor(s);
    xi.cameraLandmarks.resize(N);
    for (int i = 0; i < N; ++i)
        xi.cameraLandmarks[i] = Landmark{V3{q0[3 * i], q0[3 * i + 1], q0[3 * i + 2]}, ids[i]};
    return xi;
}
VIOGroup VIO_eqf::X() const {
    settleInvalid();
    const int N = numLandmarks();
    double s[23], g[23];
    std::vector<int> ids(N + 1);
    std::vector<double> q0(3 * N + 3), Q(5 * N + 5);
    const int rc = eqf_get_state(ctx, s, g, ids.data(), q0.data(), Q.data(), N);
    if (rc < 0)
        check(rc, "eqf_get_state");
    VIOGroup X;
    std::memcpy(X.beta.data(), g, sizeof(double) * 6);
    X.A = unpackPose(g + 6);
    X.w = V3{g[13], g[14], g[15]};
    X.B = unpackPose(g + 16);
    X.id.assign(ids.begin(), ids.begin() + N);
    X.Q.resize(N);
    for (int i = 0; i < N; ++i)
        X.Q[i] = SOT3{Qt{Q[5 * i], Q[5 * i + 1], Q[5 * i + 2], Q[5 * i + 3]}, Q[5 * i + 4]};
    return X;
}
MatrixXd VIO_eqf::Sigma() const {
    settleInvalid();
    MatrixXd S;
    S.r = S.c = 21 + 3 * numLandmarks();
    S.d.resize((size_t)S.r * S.c);
    check(eqf_get_sigma(ctx, S.d.data(), S.r), "eqf_get_sigma");
    return S;
}
void VIO_eqf::setSigma(const MatrixXd& S) { settleInvalid(); check(eqf_set_sigma(ctx, S.d.data(), S.r), "eqf_set_sigma"); }
void VIO_eqf::setSigmaDiag(const std::vector<double>& diag) { settleInvalid(); check(eqf_set_sigma_diag(ctx, diag.data(), (int)diag.size()), "eqf_set_sigma_diag"); }

bool VIO_eqf::addNewLandmarks(std::vector<Landmark>& newLandmarks, double var, bool held) { // VIO_eqf.cpp:225-245
    if (!held)
        settleInvalid();
    const int k = (int)newLandmarks.size();
    if (k == 0)
        return true;
    std::vector<int> ids(k);
    std::vector<double> p(3 * k);
    for (int i = 0; i < k; ++i) {
        ids[i] = newLandmarks[i].id;
        p[3 * i] = newLandmarks[i].p.x;
        p[3 * i + 1] = newLandmarks[i].p.y;
        p[3 * i + 2] = newLandmarks[i].p.z;
    }
    if (held) { // eqf_add_landmarks_held: refused (nothing added) when the core cannot hold them - the caller appends them behind the propagation
        const int rc = eqf_add_landmarks_held(ctx, ids.data(), p.data(), k, var);
        if (rc == EQF_E_UNSUPPORTED)
            return false;
        check(rc, "eqf_add_landmarks_held");
    } else
        check(eqf_add_landmarks(ctx, ids.data(), p.data(), k, var), "eqf_add_landmarks");
    ids_.insert(ids_.end(), ids.begin(), ids.end());
    return true;
}
void VIO_eqf::removeLandmarksByIndex(const std::vector<int>& idx) {
    settleInvalid();
    if (idx.empty())
        return;
    check(eqf_remove_landmarks(ctx, idx.data(), (int)idx.size()), "eqf_remove_landmarks");
    std::vector<char> drop(ids_.size(), 0);
    for (int i : idx)
        drop[i] = 1;
    std::vector<int> kept;
    for (size_t i = 0; i < ids_.size(); ++i)
        if (!drop[i])
            kept.push_back(ids_[i]);
    ids_ = kept;
}
// the landmarks whose id is not in the (ascending) measurement leave the state: one call of the core, which has the ids sorted (eqf_remove_unmeasured_landmarks)
bool VIO_eqf::removeUnmeasured(const std::vector<int>& measurementIds) {
    scratchIdx_.resize(ids_.size() + 1);
    int n = 0;
    const int rc = eqf_remove_unmeasured_landmarks(ctx, measurementIds.data(), (int)measurementIds.size(), scratchIdx_.data(), &n);
    if (rc == EQF_E_BAD_ARG)
        return false; // ids not ascending: the caller takes the general route
    check(rc, "eqf_remove_unmeasured_landmarks");
    if (n > 0) { // the indices are ascending: compact ids_ in place
        size_t w = 0;
        int t = 0;
        for (size_t i = 0; i < ids_.size(); ++i) {
            if (t < n && scratchIdx_[t] == (int)i) {
                ++t;
                continue;
            }
            ids_[w++] = ids_[i];
        }
        ids_.resize(w);
    }
    return true;
}
bool VIO_eqf::sameAsMapped(const std::vector<int>& measurementIds) const { return eqf_same_as_mapped(ctx, measurementIds.data(), (int)measurementIds.size()) == 1; }
bool VIO_eqf::findUnknownIds(const std::vector<int>& measurementIds, std::vector<int>& unknownJ, int& n) const {
    unknownJ.resize(measurementIds.size() + 1);
    const int rc = eqf_find_unknown_ids(ctx, measurementIds.data(), (int)measurementIds.size(), unknownJ.data(), &n);
    if (rc == EQF_E_BAD_ARG)
        return false;
    check(rc, "eqf_find_unknown_ids");
    return true;
}
void VIO_eqf::removeLandmarkByIndex(const int& idx) { removeLandmarksByIndex({idx}); } // VIO_eqf.cpp:172-178
void VIO_eqf::removeLandmarkById(const int& id) {                                      // VIO_eqf.cpp:180-186
    const auto it = std::find(ids_.begin(), ids_.end(), id);
    if (it == ids_.end())
        throw std::out_of_range("VIO_eqf::removeLandmarkById: unknown id");
    removeLandmarkByIndex((int)std::distance(ids_.begin(), it));
}
// EQF_OPT_EARLY_DOORBELL: behind an update whose lift results are not in yet the removal is deferred (settleInvalid: behind the next propagation's launch, or in front of
// whatever else looks at the landmarks first) - a landmark can be marginalised before or after the propagation, bit for bit the same for everybody else
int VIO_eqf::settleInvalid() const {
    if (!invalidPending_)
        return 0;
    invalidPending_ = false;
    const int rc = eqf_remove_invalid_at_update(ctx);
    if (rc < 0)
        check(rc, "eqf_remove_invalid_at_update");
    if (rc > 0) {
        ids_.resize(eqf_num_landmarks(ctx));
        eqf_get_ids(ctx, ids_.data(), (int)ids_.size());
    }
    return rc;
}
void VIO_eqf::removeInvalidLandmarks() { // VIO_eqf.cpp:213-223
    if (eqf_update_unsettled(ctx) == 1) {
        invalidPending_ = true;
        return;
    }
    settleInvalid();
    const int rc = eqf_remove_invalid_landmarks(ctx);
    if (rc < 0)
        check(rc, "eqf_remove_invalid_landmarks");
    if (rc > 0) {
        ids_.resize(eqf_num_landmarks(ctx));
        eqf_get_ids(ctx, ids_.data(), (int)ids_.size());
    }
}
std::array<double, 9> VIO_eqf::getLandmarkCovById(const int& id) const { // VIO_eqf.cpp:188-194 (column-major 3x3)
    settleInvalid();
    const auto it = std::find(ids_.begin(), ids_.end(), id);
    if (it == ids_.end())
        throw std::out_of_range("VIO_eqf::getLandmarkCovById: unknown id");
    const int i = (int)std::distance(ids_.begin(), it);
    std::array<double, 9> blk{};
    check(eqf_get_sigma_block(ctx, 21 + 3 * i, 21 + 3 * i, 3, 3, blk.data()), "eqf_get_sigma_block");
    return blk;
}
void VIO_eqf::integrateObserverStates(const std::vector<IMUVelocity>& imus, const std::vector<double>& dts, bool discreteLift) {
    settleInvalid();
    const int k = (int)imus.size();
    if (k == 0)
        return;
    std::vector<double> flat(13 * (size_t)k);
    for (int i = 0; i < k; ++i)
        imus[i].pack(flat.data() + 13 * i);
    check(eqf_integrate_observer(ctx, flat.data(), dts.data(), k, discreteLift ? 1 : 0), "eqf_integrate_observer");
}
void VIO_eqf::integrateObserverState(const IMUVelocity& imu, const double& dt, const bool& discreteLift) { // VIO_eqf.cpp:47-60
    settleInvalid();
    integrateObserverStates({imu}, {dt}, discreteLift);
}
void VIO_eqf::integrateRiccatiStateAccurate(const IMUVelocity& v, const double& dt, const std::array<double, 12>& Qd, const std::array<double, 8>& Pd8) {
    settleInvalid();
    double imu[13];
    v.pack(imu);
    check(eqf_integrate_riccati_accurate(ctx, imu, dt, Qd.data(), Pd8.data()), "integrateRiccatiStateAccurate");
}
void VIO_eqf::integrateRiccatiStateDiscrete(const IMUVelocity& v, const double& dt, const std::array<double, 12>& Qd, const std::array<double, 8>& Pd8) { // VIO_eqf.cpp:93-103
    settleInvalid();
    double imu[13];
    v.pack(imu);
    check(eqf_integrate_riccati_discrete(ctx, imu, dt, Qd.data(), Pd8.data()), "integrateRiccatiStateDiscrete");
}
void VIO_eqf::propagateFast(const IMUVelocity& mean, const double& dtTotal, const std::array<double, 12>& Qd, const std::array<double, 8>& Pd8,
                            const std::vector<IMUVelocity>& imus, const std::vector<double>& dts, bool discreteLift) {
    double m13[13];
    mean.pack(m13);
    std::vector<double> all(13 * imus.size());
    for (size_t i = 0; i < imus.size(); ++i)
        imus[i].pack(all.data() + 13 * i);
    check(eqf_propagate_fast(ctx, m13, dtTotal, Qd.data(), Pd8.data(), all.data(), dts.data(), (int)imus.size(), discreteLift ? 1 : 0), "eqf_propagate_fast");
}
void VIO_eqf::integrateRiccatiStateFast(const IMUVelocity& imu, const double& dt, const std::array<double, 12>& Qd, const std::array<double, 8>& Pd8) { // :62-72
    settleInvalid();
    double v[13];
    imu.pack(v);
    check(eqf_integrate_riccati_fast(ctx, v, dt, Qd.data(), Pd8.data()), "eqf_integrate_riccati_fast");
}
void VIO_eqf::performVisionUpdate(const VisionMeasurement& m, double var, const bool& useEqv, const bool& discreteCorrection) { // :105-135
    settleInvalid();
    if (m.camCoordinates.empty())
        return;
    const FlatMeas fm(m);
    const std::vector<int>& ids = fm.ids;
    const std::vector<double>& y = fm.y;
    check(eqf_vision_update(ctx, &m.cameraPtr->c, ids.data(), y.data(), (int)ids.size(), var, useEqv ? 1 : 0, discreteCorrection ? 1 : 0), "eqf_vision_update");
}
VIOState VIO_eqf::stateEstimate() const { // :137
    settleInvalid();
    const int N = numLandmarks();
    double s[23];
    std::vector<int> ids(N + 1);
    std::vector<double> p(3 * N + 3);
    const int rc = eqf_state_estimate(ctx, s, ids.data(), p.data(), N);
    if (rc < 0)
        check(rc, "eqf_state_estimate");
    VIOState xi;
    xi.sensor = unpackSensor(s);
    xi.cameraLandmarks.resize(N);
    for (int i = 0; i < N; ++i)
        xi.cameraLandmarks[i] = Landmark{V3{p[3 * i], p[3 * i + 1], p[3 * i + 2]}, ids[i]};
    return xi;
}
double VIO_eqf::computeNEES(const VIOState& trueState) const { // VIO_eqf.cpp:153-170
    settleInvalid();
    double s[23];
    packSensor(trueState.sensor, s);
    const int Nt = (int)trueState.cameraLandmarks.size();
    std::vector<int> ids(Nt + 1);
    std::vector<double> p(3 * Nt + 3);
    for (int i = 0; i < Nt; ++i) {
        ids[i] = trueState.cameraLandmarks[i].id;
        p[3 * i] = trueState.cameraLandmarks[i].p.x;
        p[3 * i + 1] = trueState.cameraLandmarks[i].p.y;
        p[3 * i + 2] = trueState.cameraLandmarks[i].p.z;
    }
    double nees = 0;
    check(eqf_compute_nees(ctx, s, ids.data(), p.data(), Nt, &nees), "eqf_compute_nees");
    return nees;
}
void VIO_eqf::outlierStats(const VisionMeasurement& m, std::vector<double>& absErr, std::vector<double>& probErr, std::vector<double>& depth2) const {
    settleInvalid();
    const int N = numLandmarks();
    absErr.assign(N, -1.0);
    probErr.assign(N, -1.0);
    depth2.assign(N, 0.0);
    if (N == 0)
        return;
    const FlatMeas fm(m);
    const std::vector<int>& ids = fm.ids;
    const std::vector<double>& y = fm.y;
    check(eqf_outlier_stats(ctx, &m.cameraPtr->c, ids.data(), y.data(), (int)ids.size(), absErr.data(), probErr.data(), depth2.data()), "eqf_outlier_stats");
}
void VIO_eqf::stageMeasurement(const VisionMeasurement& m) {
    if (m.camCoordinates.empty() || numLandmarks() == 0)
        return;
    const FlatMeas fm(m.flatHint()); // a hint: the device-side copy is checked against the (validated) measurement of the update call
    const std::vector<int>& ids = fm.ids;
    const std::vector<double>& y = fm.y;
    check(eqf_stage_measurement(ctx, ids.data(), y.data(), (int)ids.size()), "eqf_stage_measurement");
}
int VIO_eqf::statsThenUpdate(const VisionMeasurement& m, double thrAbs, double thrProb, double var, bool useEqv, bool discreteCorrection, std::vector<double>& absErr,
                              std::vector<double>& probErr, std::vector<double>& depth2, long maxOutliers) {
    settleInvalid();
    const int N = numLandmarks();
    absErr.assign(N, -1.0);
    probErr.assign(N, -1.0);
    depth2.assign(N, 0.0);
    const FlatMeas fm(m);
    const std::vector<int>& ids = fm.ids;
    const std::vector<double>& y = fm.y;
    int updated = 0;
    if (maxOutliers < 0) {
        check(eqf_stats_then_update(ctx, &m.cameraPtr->c, ids.data(), y.data(), (int)ids.size(), thrAbs, thrProb, var, useEqv ? 1 : 0, discreteCorrection ? 1 : 0,
                                    absErr.data(), probErr.data(), depth2.data(), &updated),
              "eqf_stats_then_update");
        return updated;
    }
    // removeOutliers' decision may be taken on the device (eqf_hip.h): the landmarks it discarded are gone from the state when the call returns
    std::vector<int> removed(N + 1);
    int nRemoved = 0;
    check(eqf_stats_select_update(ctx, &m.cameraPtr->c, ids.data(), y.data(), (int)ids.size(), thrAbs, thrProb, (int)std::min<long>(maxOutliers, 1 << 30), var, useEqv ? 1 : 0,
                                  discreteCorrection ? 1 : 0, absErr.data(), probErr.data(), depth2.data(), &updated, removed.data(), &nRemoved),
          "eqf_stats_select_update");
    if (nRemoved > 0) {
        std::vector<char> drop(ids_.size(), 0);
        for (int t = 0; t < nRemoved; ++t)
            drop[removed[t]] = 1;
        std::vector<int> kept;
        for (size_t i = 0; i < ids_.size(); ++i)
            if (!drop[i])
                kept.push_back(ids_[i]);
        ids_ = kept;
    }
    return updated;
}

// ---------------------------------------------------------------- Settings (VIOFilterSettings.h)
VIOFilter::Settings::Settings(const eqvio_settings& s) {
    biasOmegaProcessVariance = s.biasOmegaProcessVariance;
    biasAccelProcessVariance = s.biasAccelProcessVariance;
    attitudeProcessVariance = s.attitudeProcessVariance;
    positionProcessVariance = s.positionProcessVariance;
    velocityProcessVariance = s.velocityProcessVariance;
    cameraAttitudeProcessVariance = s.cameraAttitudeProcessVariance;
    cameraPositionProcessVariance = s.cameraPositionProcessVariance;
    pointProcessVariance = s.pointProcessVariance;
    velGyrNoise = s.velGyrNoise;
    velAccNoise = s.velAccNoise;
    velGyrBiasWalk = s.velGyrBiasWalk;
    velAccBiasWalk = s.velAccBiasWalk;
    measurementNoise = s.measurementNoise;
    outlierThresholdAbs = s.outlierThresholdAbs;
    outlierThresholdProb = s.outlierThresholdProb;
    featureRetention = s.featureRetention;
    initialAttitudeVariance = s.initialAttitudeVariance;
    initialPositionVariance = s.initialPositionVariance;
    initialVelocityVariance = s.initialVelocityVariance;
    initialCameraAttitudeVariance = s.initialCameraAttitudeVariance;
    initialCameraPositionVariance = s.initialCameraPositionVariance;
    initialPointVariance = s.initialPointVariance;
    initialPointDepthVariance = s.initialPointDepthVariance;
    initialBiasOmegaVariance = s.initialBiasOmegaVariance;
    initialBiasAccelVariance = s.initialBiasAccelVariance;
    initialSceneDepth = s.initialSceneDepth;
    useDiscreteInnovationLift = s.useDiscreteInnovationLift != 0;
    useDiscreteVelocityLift = s.useDiscreteVelocityLift != 0;
    useDiscreteStateMatrix = s.useDiscreteStateMatrix != 0;
    fastRiccati = s.fastRiccati != 0;
    useMedianDepth = s.useMedianDepth != 0;
    useFeaturePredictions = s.useFeaturePredictions != 0;
    useEquivariantOutput = s.useEquivariantOutput != 0;
    removeLostLandmarks = s.removeLostLandmarks != 0;
    if (s.coordinateChoice < 0 || s.coordinateChoice > 2) // coordinateSelection, VIOFilterSettings.h:33-46
        throw std::runtime_error("Invalid coordinate choice. Valid choices are Euclidean, InvDepth, Normal.");
    coordinateChoice = (CoordinateChoice)s.coordinateChoice;
    cameraOffset = unpackPose(s.cameraOffset);
}
std::vector<double> VIOFilter::Settings::constructInitialStateCovarianceDiag(const size_t& N) const { // :208-229
    std::vector<double> d(21 + 3 * N, initialPointVariance);
    const double v[7] = {initialBiasOmegaVariance, initialBiasAccelVariance, initialAttitudeVariance, initialPositionVariance,
                         initialVelocityVariance,  initialCameraAttitudeVariance, initialCameraPositionVariance};
    for (int b = 0; b < 7; ++b)
        for (int k = 0; k < 3; ++k)
            d[3 * b + k] = v[b];
    if (initialPointDepthVariance > 0)
        for (size_t i = 0; i < N; ++i)
            d[21 + 3 * i + 2] = initialPointDepthVariance;
    return d;
}
std::array<double, 8> VIOFilter::Settings::constructStateGainDiag8() const { // :176-190
    return {biasOmegaProcessVariance, biasAccelProcessVariance,      attitudeProcessVariance,       positionProcessVariance,
            velocityProcessVariance,  cameraAttitudeProcessVariance, cameraPositionProcessVariance, pointProcessVariance};
}
std::array<double, 12> VIOFilter::Settings::constructInputGainDiag() const { // :192-201
    std::array<double, 12> q{};
    const double v[4] = {velGyrNoise * velGyrNoise, velAccNoise * velAccNoise, velGyrBiasWalk * velGyrBiasWalk, velAccBiasWalk * velAccBiasWalk};
    for (int b = 0; b < 4; ++b)
        for (int k = 0; k < 3; ++k)
            q[3 * b + k] = v[b];
    return q;
}

// ---------------------------------------------------------------- VIOFilter (src/VIOFilter.cpp)
VIOFilter::~VIOFilter() = default;

VIOFilter::VIOFilter(const VIOFilter::Settings& s) { // :31-41
    settings = std::make_unique<VIOFilter::Settings>(s);
    filterState.create(s.device, s.maxLandmarks, s.coordinateChoice);
    VIOState xi0;
    xi0.sensor.cameraOffset = s.cameraOffset;
    filterState.set(xi0, VIOGroup());
    filterState.setSigmaDiag(settings->constructInitialStateCovarianceDiag());
}
VIOFilter::VIOFilter(const VIOState& xi0, const VIOFilter::Settings& s, const double& time) { // :43-56
    settings = std::make_unique<VIOFilter::Settings>(s);
    filterState.create(s.device, std::max<int>(s.maxLandmarks, (int)xi0.cameraLandmarks.size()), s.coordinateChoice);
    VIOGroup X;
    for (const Landmark& lm : xi0.cameraLandmarks) {
        X.Q.emplace_back(SOT3());
        X.id.emplace_back(lm.id);
    }
    filterState.set(xi0, X);
    filterState.setSigmaDiag(settings->constructInitialStateCovarianceDiag(xi0.cameraLandmarks.size()));
    filterState.currentTime = time;
    initialisedFlag = true;
}
void VIOFilter::processIMUData(const IMUVelocity& imu) { // :58-63
    if (!initialisedFlag)
        initialiseFromIMUData(imu);
    velocityBuffer.emplace_back(imu);
}
void VIOFilter::initialiseFromIMUData(const IMUVelocity& imu) { // :65-78
    VIOState xi0 = filterState.xi0();
    const VIOGroup X = filterState.X();
    xi0.sensor.inputBias.fill(0.0);
    xi0.sensor.pose = pose_identity();
    xi0.sensor.velocity = V3{0, 0, 0};
    initialisedFlag = true;
    xi0.sensor.pose.R = so3_from_vectors(normalized(imu.acc), V3{0, 0, 1});
    filterState.set(xi0, X);
    filterState.currentTime = imu.stamp;
}
void VIOFilter::setState(const VIOState& xi) { // :80-92
    VIOGroup X;
    X.id = xi.getIds();
    X.Q.assign(X.id.size(), SOT3());
    filterState.set(xi, X);
    const int N = (int)xi.cameraLandmarks.size();
    std::vector<double> d = settings->constructInitialStateCovarianceDiag(0);
    d.resize(21 + 3 * N, 1.0 * settings->initialPointVariance);
    filterState.setSigmaDiag(d);
    initialisedFlag = true;
}
void VIOFilter::setLandmarks(const std::vector<Landmark>& lms) { // :94-110
    // The reference overwrites the landmark block of Sigma in place and keeps the sensor block and the cross terms
    // (which requires the landmark count to be unchanged). Same here: read Sigma back, patch, write.
    MatrixXd S = filterState.Sigma();
    const std::vector<double> full = settings->constructInitialStateCovarianceDiag(lms.size());
    const int k = 3 * (int)lms.size();
    if (21 + k != S.r)
        throw std::invalid_argument("VIOFilter::setLandmarks: landmark count differs from the filter state");
    for (int j = 0; j < k; ++j)
        for (int i = 0; i < k; ++i)
            S(21 + i, 21 + j) = (i == j) ? full[21 + i] : 0.0;
    VIOState xi0 = filterState.xi0();
    VIOGroup X = filterState.X();
    xi0.cameraLandmarks = lms;
    X.Q.assign(lms.size(), SOT3());
    X.id.clear();
    for (const Landmark& lm : lms)
        X.id.emplace_back(lm.id);
    filterState.set(xi0, X);
    filterState.setSigma(S);
}
void VIOFilter::augmentLandmarkStates(const std::vector<int>& newIds, const VIOState& provided) { // :112-132
    removeOldLandmarks(newIds);
    std::vector<Landmark> newLandmarks;
    const std::vector<int>& have = filterState.ids();
    for (const int& id : newIds) {
        if (std::find(have.begin(), have.end(), id) != have.end())
            continue;
        const auto it2 = std::find_if(provided.cameraLandmarks.begin(), provided.cameraLandmarks.end(), [&id](const Landmark& lm) { return lm.id == id; });
        if (it2 == provided.cameraLandmarks.end())
            throw std::out_of_range("augmentLandmarkStates: id missing from the provided state");
        newLandmarks.emplace_back(*it2);
    }
    filterState.addNewLandmarks(newLandmarks, settings->initialPointVariance);
}
bool VIOFilter::integrateUpToTime(const double& newTime) { // :134-192
    if (newTime <= filterState.currentTime || filterState.currentTime < 0 || velocityBuffer.empty())
        return false;
    std::vector<double> dts(velocityBuffer.size());
    for (size_t i = 0; i < velocityBuffer.size(); ++i) {
        const double t0 = std::max(velocityBuffer.at(i).stamp, filterState.currentTime);
        const double t1 = i + 1 < velocityBuffer.size() ? std::min(velocityBuffer.at(i + 1).stamp, newTime) : newTime;
        dts[i] = std::max(t1 - t0, 0.0);
    }
    if (settings->fastRiccati) {
        double accumulatedTime = 0;
        IMUVelocity accumulatedVelocity = IMUVelocity::Zero();
        for (size_t i = 0; i < velocityBuffer.size(); ++i) {
            accumulatedTime += dts[i];
            accumulatedVelocity = accumulatedVelocity + velocityBuffer.at(i) * dts[i];
        }
        accumulatedVelocity = accumulatedVelocity * (1.0 / accumulatedTime);
        // The observer steps do not depend on the Riccati state (VIOFilter.cpp:138): the Riccati step at the current X and all
        // observer steps go to the device in one call, the observer's landmark kernel queued ahead of the Sigma propagation.
        filterState.propagateFast(accumulatedVelocity, accumulatedTime, settings->constructInputGainDiag(), settings->constructStateGainDiag8(), velocityBuffer, dts,
                                  settings->useDiscreteVelocityLift);
    } else {
        // VIOFilter.cpp:160-178: per IMU sample, the discrete-A or the accurate Riccati step at the current X, then the observer step
        for (size_t i = 0; i < velocityBuffer.size(); ++i) {
            if (dts[i] > 0) {
                if (settings->useDiscreteStateMatrix)
                    filterState.integrateRiccatiStateDiscrete(velocityBuffer.at(i), dts[i], settings->constructInputGainDiag(), settings->constructStateGainDiag8());
                else
                    filterState.integrateRiccatiStateAccurate(velocityBuffer.at(i), dts[i], settings->constructInputGainDiag(), settings->constructStateGainDiag8());
            }
            filterState.integrateObserverState(velocityBuffer.at(i), dts[i], settings->useDiscreteVelocityLift);
        }
    }
    filterState.currentTime = newTime;
    auto it = std::find_if(velocityBuffer.begin(), velocityBuffer.end(), [this](const IMUVelocity& v) { return v.stamp >= this->filterState.currentTime; });
    if (it != velocityBuffer.begin()) {
        --it;
        velocityBuffer.erase(velocityBuffer.begin(), it);
    }
    return true;
}
void VIOFilter::processVisionData(const VisionMeasurement& measurement) { // :194-241
    HP_SCOPE("processVisionData");
    // The flat arrays of the measurement are validated against its std::map ONCE per call (Validated) - in front of the propagation when the frame gains or loses
    // landmarks, behind its launch when it does not: in the steady frame the unvalidated cache (flatHint) holds exactly the ids the last update mapped, one per landmark,
    // so nothing is removed or added on its word; the walk (1 us at 200 features, GPU idle time in front of the launch) then runs beside the propagation kernel, and if
    // it finds the map edited the lost / new landmarks are dealt with THERE, behind the propagation - the reference's own order (:210-217 behind :196).
    std::optional<VisionMeasurement::Validated> oneWalk;
    loopTimer.startTiming("propagation");
    // Round 5: the landmarks that are not in this measurement leave the state BEFORE the propagation instead of behind it (reference: :210-212 behind :196). The
    // propagation is block triangular - a landmark's rows and columns depend on the sensor block and on themselves only - so marginalising a landmark out before or
    // after it gives the other entries bit for bit, and the order matters to the clock: the removal is recorded on the host (7.6 us at 200 landmarks) and applied by the
    // pass in front of the propagation kernel, which used to run while the GPU sat idle behind that kernel; now the host's share (this, and the new landmarks below)
    // overlaps with kernels. Only when the propagation will really take place (integrateUpToTime's own precondition, :135): a frame it skips must not lose landmarks.
    const bool willIntegrate = !(measurement.stamp <= filterState.currentTime || filterState.currentTime < 0 || velocityBuffer.empty());
    bool quiet = false; // the cache says: the ids of the last mapped measurement, one per landmark - nobody lost, nobody new
    if (initialisedFlag && willIntegrate && settings->fastRiccati)
        quiet = filterState.sameAsMapped(*measurement.flatHint().first);
    if (!quiet)
        oneWalk.emplace(measurement);
    bool removedEarly = initialisedFlag && willIntegrate && settings->removeLostLandmarks;
    if (removedEarly && !quiet) {
        HP_SCOPE("pv.removeOldLandmarks");
        removeOldLandmarks(measurement.flatIds());
    }
    // ... and the frame's NEW landmarks enter in front of the propagation too, held (eqf_add_landmarks_held): with a fixed initial depth (both shipped dataset
    // configurations) a new landmark depends on its pixel only (a point in the camera frame AT the measurement's time: nothing of the propagation enters), and the
    // propagation passes a held landmark through untouched - the same state as appending it behind the propagation (:217), bit for bit. Every id of the measurement
    // is then known before the propagation: the measurement is staged, the propagation kernel evaluates its output blocks and creates the new landmarks itself, and a
    // frame with landmark turnover is three launches like any other. Refused by the core (options, capacity): they are appended behind the propagation as before.
    bool heldAdd = quiet; // (a quiet frame has no new landmark)
    if (!quiet && initialisedFlag && willIntegrate && settings->fastRiccati && !settings->useMedianDepth && filterState.holdSupported()) {
        HP_SCOPE("pv.addNewLandmarks");
        heldAdd = addNewLandmarks(measurement, nullptr, true);
    }
    // The measurement is in hand before the propagation (VIOFilter.cpp:194-196): hand it to the device now, so that it travels to
    // HBM inside the propagation kernel instead of across PCIe in the update's first kernel (a hint: ignored if an id is unknown).
    if (initialisedFlag && settings->fastRiccati) {
        HP_SCOPE("pv.stageMeasurement");
        filterState.stageMeasurement(measurement);
    }
    bool integrationFlag;
    {
        HP_SCOPE("pv.integrateUpToTime");
        integrationFlag = integrateUpToTime(measurement.stamp);
    }
    if (quiet) { // the walk the decisions above stand on, beside the propagation kernel
        const size_t before = measurement.flatRebuilds();
        oneWalk.emplace(measurement);
        if (measurement.flatRebuilds() != before) // the map had been edited behind the cache: lost and new landmarks are looked for now, behind the propagation
            removedEarly = false, heldAdd = false;
    }
    // (EQF_OPT_EARLY_DOORBELL) the invalid landmarks of the previous update leave now, beside the propagation kernel; if there were any, the measurement may hold their ids
    // again - new landmarks, as for the reference, whose removal ran before this call: they are looked for below
    if (filterState.settleInvalid() > 0)
        heldAdd = false;
    if (!integrationFlag || !initialisedFlag)
        return;
    loopTimer.endTiming("propagation");

    loopTimer.startTiming("preprocessing");
    if (settings->removeLostLandmarks && !removedEarly) {
        HP_SCOPE("pv.removeOldLandmarks");
        removeOldLandmarks(measurement.flatIds());
    }
    // With a fixed initial depth (both shipped dataset configurations) a new landmark depends on its pixel only, and the outlier test
    // never looks at it (it is not in the state yet in the reference's order; here its residual is zero by construction): appending the
    // new landmarks BEFORE the test instead of after it gives the same state - removing outliers afterwards only compacts the older rows,
    // the new rows keep their relative order at the end - and lets a frame with landmark turnover take the one-round-trip path too
    // (statistics + update queued back to back) instead of statistics -> host -> append -> update.
    const bool earlyAdd = !settings->useMedianDepth;
    if (earlyAdd && !heldAdd) {
        HP_SCOPE("pv.addNewLandmarks");
        addNewLandmarks(measurement, nullptr);
    }
    std::vector<double> depth2;
    // Every measured id already in the state (no landmark to add) and something to update: queue the outlier statistics and
    // the update back to back (eqf_stats_then_update). If a measured landmark exceeds a threshold the device cancels the
    // update and the frame continues below exactly as the reference does, with the statistics already in hand; if some id
    // is unknown nothing is computed (-1). The copy of the measurement the reference makes (VIOFilter.cpp:213) is only
    // needed when outliers may be erased from it, i.e. on the path below.
    bool haveStats = false;
    std::vector<double> absErr, probErr;
    if (!measurement.camCoordinates.empty() && filterState.numLandmarks() > 0) {
        // With the new landmarks already in the state (earlyAdd) nothing of the frame needs the host between the statistics and the update, so
        // the outlier decision itself may run on the device (maxOutliers as in removeOutliers, VIOFilter.cpp:305); otherwise the decision stays here.
        HP_SCOPE("pv.statsThenUpdate");
        const long maxOutliers = earlyAdd ? (long)(size_t)((1.0 - settings->featureRetention) * measurement.camCoordinates.size()) : -1;
        const int r = filterState.statsThenUpdate(measurement, settings->outlierThresholdAbs, settings->outlierThresholdProb, settings->constructOutputGainVar(),
                                                  settings->useEquivariantOutput, settings->useDiscreteInnovationLift, absErr, probErr, depth2, maxOutliers);
        if (r == 1) {
            loopTimer.endTiming("preprocessing");
            loopTimer.startTiming("correction");
            filterState.removeInvalidLandmarks();
            loopTimer.endTiming("correction");
            return;
        }
        haveStats = (r == 0);
    }
    VisionMeasurement matchedMeasurement = measurement;
    removeOutliers(matchedMeasurement, depth2, haveStats ? &absErr : nullptr, haveStats ? &probErr : nullptr);
    if (!earlyAdd)
        addNewLandmarks(matchedMeasurement, &depth2);
    loopTimer.endTiming("preprocessing");

    if (matchedMeasurement.camCoordinates.empty())
        return;

    loopTimer.startTiming("correction");
    if (!settings->removeLostLandmarks) {
        // the reference tolerates state landmarks without a measurement (zero block columns of C,
        // EqFMatrices.cpp:61-78) but not measurements without a landmark; same contract here.
    }
    filterState.performVisionUpdate(matchedMeasurement, settings->constructOutputGainVar(), settings->useEquivariantOutput, settings->useDiscreteInnovationLift);
    filterState.removeInvalidLandmarks();
    loopTimer.endTiming("correction");
}
VIOState VIOFilter::stateEstimate() const { return filterState.stateEstimate(); }
const VIO_eqf& VIOFilter::viewEqFState() const { return filterState; }
double VIOFilter::getTime() const { return filterState.currentTime; }
VIOState integrateSystemFunction(const VIOState& state, const IMUVelocity& velocity, const double& dt) { // VIOState.cpp:28-68
    VIOState ns;
    const VIOSensorState& s = state.sensor;
    const V3 gyr = velocity.gyr - V3{s.inputBias[0], s.inputBias[1], s.inputBias[2]}; // v_est = velocity - bias (IMUVelocity.cpp:52-58)
    const V3 acc = velocity.acc - V3{s.inputBias[3], s.inputBias[4], s.inputBias[5]};
    const V3 dbg = dt * velocity.gyrBiasVel, dba = dt * velocity.accBiasVel;
    ns.sensor.inputBias = {s.inputBias[0] + dbg.x, s.inputBias[1] + dbg.y, s.inputBias[2] + dbg.z, s.inputBias[3] + dba.x, s.inputBias[4] + dba.y, s.inputBias[5] + dba.z};
    const V3 gravity{0, 0, -GRAVITY_CONSTANT_};
    Pose poseChange;
    poseChange.R = eqf::so3_exp(dt * gyr);
    const V3 inertialStep = dt * eqf::q_rot(s.pose.R, s.velocity) + (0.5 * dt * dt) * (eqf::q_rot(s.pose.R, acc) + gravity);
    poseChange.x = eqf::q_rot(eqf::q_inv(s.pose.R), inertialStep);
    ns.sensor.pose = eqf::pose_mul(s.pose, poseChange);
    const V3 inertialVelocityDiff = eqf::q_rot(s.pose.R, acc) + gravity;
    ns.sensor.velocity = eqf::q_rot(eqf::q_inv(ns.sensor.pose.R), eqf::q_rot(s.pose.R, s.velocity) + dt * inertialVelocityDiff);
    const Pose cameraPoseChangeInv = eqf::pose_mul(eqf::pose_mul(eqf::pose_inv(s.cameraOffset), eqf::pose_inv(poseChange)), s.cameraOffset);
    ns.cameraLandmarks.resize(state.cameraLandmarks.size());
    for (size_t i = 0; i < state.cameraLandmarks.size(); ++i)
        ns.cameraLandmarks[i] = Landmark{eqf::pose_act(cameraPoseChangeInv, state.cameraLandmarks[i].p), state.cameraLandmarks[i].id};
    ns.sensor.cameraOffset = s.cameraOffset;
    return ns;
}
VIOState VIO_eqf::predictState(const double& stamp, const std::vector<IMUVelocity>& imuVelocities) const { // VIO_eqf.cpp:139-151
    VIOState statePrediction = stateEstimate();
    for (size_t i = 0; i < imuVelocities.size(); ++i) {
        const double t0 = std::max(imuVelocities.at(i).stamp, this->currentTime);
        const double t1 = i + 1 < imuVelocities.size() ? std::min(imuVelocities.at(i + 1).stamp, stamp) : stamp;
        const double dt = std::max(t1 - t0, 0.0);
        statePrediction = integrateSystemFunction(statePrediction, imuVelocities.at(i), dt);
    }
    return statePrediction;
}
VisionMeasurement VIOFilter::getFeaturePredictions(const GICameraPtr& camPtr, const double& stamp) { // :247-252
    VisionMeasurement r;
    if (settings->useFeaturePredictions) { // measureSystemState(predictState(...)), VIOState.cpp:70-78
        const VIOState pred = filterState.predictState(stamp, velocityBuffer);
        for (const Landmark& lm : pred.cameraLandmarks) {
            double u, v;
            camPtr->projectPoint(lm.p, u, v);
            r.camCoordinates[lm.id] = {u, v};
        }
        r.cameraPtr = camPtr;
    }
    return r;
}
bool VIOFilter::addNewLandmarks(const VisionMeasurement& measurement, const std::vector<double>* depth2, bool held) { // :258-278
    std::vector<Landmark> newLandmarks;
    {
        // ascending measurement ids (a VisionMeasurement's): the core, which keeps the state's ids sorted, says which of them are new in one merge pass
        const auto fv = measurement.flat();
        const std::vector<int>& mids = *fv.first;
        const std::vector<double>& my = *fv.second;
        int n = 0;
        if (filterState.findUnknownIds(mids, unknownScratch_, n)) {
            if (n == 0)
                return true;
            newLandmarks.reserve(n);
            for (int t = 0; t < n; ++t) {
                const int j = unknownScratch_[t];
                const V3 bearing = measurement.cameraPtr->undistortPoint(my[2 * j], my[2 * j + 1]);
                newLandmarks.emplace_back(Landmark{bearing, mids[j]});
            }
            const double initialDepth = settings->useMedianDepth ? getMedianSceneDepth(depth2) : settings->initialSceneDepth;
            for (Landmark& blm : newLandmarks)
                blm.p = initialDepth * blm.p;
            return filterState.addNewLandmarks(newLandmarks, settings->initialPointVariance, held);
        }
    }
    if (held)
        return false; // (ids not ascending: the general route below runs behind the propagation)
    // O(M log N) membership instead of the reference's O(M N) scan. The state's ids are usually ascending already (a tracker numbers its features
    // as they appear, removals keep the order): no copy and no sort then, and one merge pass against the measurement's ascending ids.
    const std::vector<int>& stateIds = filterState.ids();
    std::vector<int> sortedCopy;
    if (!std::is_sorted(stateIds.begin(), stateIds.end())) {
        sortedCopy = stateIds;
        std::sort(sortedCopy.begin(), sortedCopy.end());
    }
    const std::vector<int>& have = sortedCopy.empty() ? stateIds : sortedCopy;
    const auto flatView = measurement.flat();
    const std::vector<int>& mids = *flatView.first;
    const std::vector<double>& my = *flatView.second;
    const bool merge = std::is_sorted(mids.begin(), mids.end());
    size_t h = 0;
    for (size_t j = 0; j < mids.size(); ++j) {
        const int ccId = mids[j];
        bool known;
        if (merge) {
            while (h < have.size() && have[h] < ccId)
                ++h;
            known = h < have.size() && have[h] == ccId;
        } else
            known = std::binary_search(have.begin(), have.end(), ccId);
        if (!known) {
            const V3 bearing = measurement.cameraPtr->undistortPoint(my[2 * j], my[2 * j + 1]);
            newLandmarks.emplace_back(Landmark{bearing, ccId});
        }
    }
    if (newLandmarks.empty())
        return true;
    const double initialDepth = settings->useMedianDepth ? getMedianSceneDepth(depth2) : settings->initialSceneDepth;
    for (Landmark& blm : newLandmarks)
        blm.p = initialDepth * blm.p;
    return filterState.addNewLandmarks(newLandmarks, settings->initialPointVariance);
}
void VIOFilter::removeOldLandmarks(const std::vector<int>& measurementIds) { // :280-302
    if (filterState.removeUnmeasured(measurementIds)) // (ascending measurement ids, as a VisionMeasurement's are: one merge pass inside the core)
        return;
    const std::vector<int>& have = filterState.ids();
    std::vector<int> lost;
    bool bothSorted;
    {
        HP_SCOPE("ro.sorted");
        bothSorted = std::is_sorted(measurementIds.begin(), measurementIds.end()) && std::is_sorted(have.begin(), have.end());
    }
    const bool sorted = bothSorted || std::is_sorted(measurementIds.begin(), measurementIds.end()); // ids from a VisionMeasurement are
    if (bothSorted) { // both ascending (the usual case): one merge pass
        {
            HP_SCOPE("ro.merge");
            size_t q = 0;
            for (int i = 0; i < (int)have.size(); ++i) {
                while (q < measurementIds.size() && measurementIds[q] < have[i])
                    ++q;
                if (q == measurementIds.size() || measurementIds[q] != have[i])
                    lost.push_back(i);
            }
        }
        HP_SCOPE("ro.removeByIndex");
        filterState.removeLandmarksByIndex(lost);
        return;
    }
    for (int i = 0; i < (int)have.size(); ++i) {
        const bool found = sorted ? std::binary_search(measurementIds.begin(), measurementIds.end(), have[i])
                                  : std::find(measurementIds.begin(), measurementIds.end(), have[i]) != measurementIds.end();
        if (!found)
            lost.push_back(i);
    }
    filterState.removeLandmarksByIndex(lost); // one compaction pass instead of one per landmark
}
void VIOFilter::removeOutliers(VisionMeasurement& measurement, std::vector<double>& depth2, const std::vector<double>* absErrIn,
                               const std::vector<double>* probErrIn) { // :304-364
    const size_t maxOutliers = (size_t)((1.0 - settings->featureRetention) * measurement.camCoordinates.size());
    std::vector<double> absErr, probErr;
    if (absErrIn && probErrIn) { // statistics of a cancelled speculative tail (depth2 filled by the same call)
        absErr = *absErrIn;
        probErr = *probErrIn;
    } else {
        filterState.outlierStats(measurement, absErr, probErr, depth2);
    }
    const std::vector<int>& ids = filterState.ids();
    std::vector<int> proposedOutliers;
    std::map<int, double> absoluteOutliers, probabilisticOutliers;
    // the reference iterates yHat (a std::map) in ascending id order for both passes
    std::vector<int> order(ids.size());
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&ids](int a, int b) { return ids[a] < ids[b]; });
    for (const int i : order) {
        if (absErr[i] < 0)
            continue; // not measured
        if (absErr[i] > settings->outlierThresholdAbs) {
            absoluteOutliers[ids[i]] = absErr[i];
            proposedOutliers.emplace_back(ids[i]);
        }
    }
    for (const int i : order) {
        if (absErr[i] < 0 || absoluteOutliers.count(ids[i]))
            continue;
        if (probErr[i] > settings->outlierThresholdProb) {
            probabilisticOutliers[ids[i]] = probErr[i];
            proposedOutliers.emplace_back(ids[i]);
        }
    }
    std::sort(proposedOutliers.begin(), proposedOutliers.end(), [&absoluteOutliers, &probabilisticOutliers](const int& a, const int& b) {
        if (absoluteOutliers.count(a)) {
            if (absoluteOutliers.count(b))
                return absoluteOutliers.at(a) < absoluteOutliers.at(b);
            return false;
        }
        if (absoluteOutliers.count(b))
            return true;
        return probabilisticOutliers.at(a) < probabilisticOutliers.at(b);
    });
    std::reverse(proposedOutliers.begin(), proposedOutliers.end());
    if (proposedOutliers.size() > maxOutliers)
        proposedOutliers.erase(proposedOutliers.begin() + maxOutliers, proposedOutliers.end());
    if (proposedOutliers.empty())
        return;
    std::vector<int> idx;
    for (const int lmId : proposedOutliers) {
        idx.push_back((int)std::distance(ids.begin(), std::find(ids.begin(), ids.end(), lmId)));
        measurement.camCoordinates.erase(lmId);
    }
    // keep depth2 aligned with the compacted state
    std::vector<char> drop(ids.size(), 0);
    for (int i : idx)
        drop[i] = 1;
    std::vector<double> d2;
    for (size_t i = 0; i < drop.size(); ++i)
        if (!drop[i])
            d2.push_back(depth2[i]);
    depth2 = d2;
    filterState.removeLandmarksByIndex(idx);
}
double VIOFilter::getMedianSceneDepth(const std::vector<double>* depth2) const { // :366-380
    std::vector<double> depthsSquared;
    if (depth2) {
        depthsSquared = *depth2;
    } else {
        const VIOState est = stateEstimate();
        for (const Landmark& lm : est.cameraLandmarks)
            depthsSquared.push_back(norm2(lm.p));
    }
    const auto midway = depthsSquared.begin() + depthsSquared.size() / 2;
    std::nth_element(depthsSquared.begin(), midway, depthsSquared.end());
    double medianDepth = settings->initialSceneDepth;
    if (!(midway == depthsSquared.end()))
        medianDepth = std::pow(*midway, 0.5);
    return medianDepth;
}

} // namespace eqvio_amd
