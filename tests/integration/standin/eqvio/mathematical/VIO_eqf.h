// TEST SCAFFOLDING. The value types of the reference's EqF interface re-declared with the reference's type, member and function
// NAMES and argument lists (include/eqvio/mathematical/{VIOState,VIOGroup,IMUVelocity,VisionMeasurement,VIO_eqf}.h), only as far as the
// binding in tests/integration/VIO_eqf_mi355x.cpp touches them, over the stand-in Eigen / LiePP / GIFT headers next to this file.
// The two marked places are the ONLY edits the binding needs in the reference's real VIO_eqf.h.
#pragma once
#include <map>
#include <memory>
#include <vector>

#include "Eigen/Dense"
#include "GIFT/Camera.h"
#include "liepp/SE3.h"

struct Landmark {
    Eigen::Vector3d p;
    int id = -1;
    constexpr static int CompDim = 3;
};
struct VIOSensorState {
    Eigen::Matrix<double, 6, 1> inputBias;
    liepp::SE3d pose;
    Eigen::Vector3d velocity;
    liepp::SE3d cameraOffset;
    constexpr static int CompDim = 6 + 6 + 3 + 6;
};
struct VIOState {
    VIOSensorState sensor;
    std::vector<Landmark> cameraLandmarks;
    std::vector<int> getIds() const {
        std::vector<int> ids;
        for (const Landmark& lm : cameraLandmarks) ids.push_back(lm.id);
        return ids;
    }
};
struct VIOGroup {
    Eigen::Matrix<double, 6, 1> beta;
    liepp::SE3d A;
    Eigen::Vector3d w;
    liepp::SE3d B;
    std::vector<liepp::SOT3d> Q;
    std::vector<int> id;
    static VIOGroup Identity(const std::vector<int>& ids = {}) {
        VIOGroup X;
        X.id = ids;
        X.Q.resize(ids.size());
        return X;
    }
};
struct IMUVelocity {
    double stamp = 0;
    Eigen::Vector3d gyr, acc;
    Eigen::Vector3d gyrBiasVel = Eigen::Vector3d::Zero(), accBiasVel = Eigen::Vector3d::Zero();
    constexpr static int CompDim = 12;
    static IMUVelocity Zero() { return IMUVelocity(); }
    IMUVelocity operator+(const IMUVelocity& o) const { // src/mathematical/IMUVelocity.cpp:42-50
        IMUVelocity r;
        r.stamp = (stamp > 0) ? stamp : o.stamp;
        r.gyr = gyr + o.gyr, r.acc = acc + o.acc, r.gyrBiasVel = gyrBiasVel + o.gyrBiasVel, r.accBiasVel = accBiasVel + o.accBiasVel;
        return r;
    }
    IMUVelocity operator*(const double& c) const {
        IMUVelocity r;
        r.stamp = stamp;
        r.gyr = gyr * c, r.acc = acc * c, r.gyrBiasVel = gyrBiasVel * c, r.accBiasVel = accBiasVel * c;
        return r;
    }
};
struct VisionMeasurement {
    double stamp = 0;
    std::map<int, Eigen::Vector2d> camCoordinates;
    GIFT::GICameraPtr cameraPtr;
    std::vector<int> getIds() const {
        std::vector<int> ids;
        for (const auto& kv : camCoordinates) ids.push_back(kv.first);
        return ids;
    }
};
inline VisionMeasurement operator-(const VisionMeasurement& y1, const VisionMeasurement& y2) { // VisionMeasurement.cpp:58-69: the ids both carry
    VisionMeasurement d;
    d.stamp = y1.stamp;
    d.cameraPtr = y1.cameraPtr;
    for (const auto& kv : y1.camCoordinates) {
        const auto it2 = y2.camCoordinates.find(kv.first);
        if (it2 != y2.camCoordinates.end()) d.camCoordinates[kv.first] = kv.second - it2->second;
    }
    return d;
}
inline VisionMeasurement measureSystemState(const VIOState& xi, const GIFT::GICameraPtr& cam) { // VIOState.cpp:70-78
    VisionMeasurement y;
    y.cameraPtr = cam;
    for (const Landmark& lm : xi.cameraLandmarks) y.camCoordinates[lm.id] = cam->projectPoint(lm.p);
    return y;
}
struct EqFCoordinateSuite {}; // the reference's struct of function objects; the binding only compares addresses
extern const EqFCoordinateSuite EqFCoordinateSuite_euclid, EqFCoordinateSuite_invdepth, EqFCoordinateSuite_normal;

// ---- ADDED for the MI355X binding (1 of 2): the device-side twin of one VIO_eqf. Copying a filter clones the twin.
struct eqf_ctx; // include/eqf_hip.h
namespace eqvio_mi355x {
struct DeviceTwin {
    eqf_ctx* ctx = nullptr;
    int chart = -1, capacity = 0;
    bool deviceNewer = false; // Sigma / X / xi0 numbers on the device are ahead of the host members
    bool hostEdited = true;   // the host members were assigned (construction, VIOFilter::setState, ...): upload before the next call
    // getOutputCovById: the reference's removeOutliers asks for one landmark at a time (VIOFilter.cpp:329); the first call after the state changed fetches
    // the 2 x 2 output covariances of ALL landmarks in one device call (eqf_output_cov_all), the others are served from here. Not copied with the twin.
    std::vector<double> outCov;
    bool outCovValid = false;
    size_t outCovLast = (size_t)-1; // index of the landmark the last getOutputCovById asked for
    double outCovCam[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; // the camera the cache was computed for
    void touch() { deviceNewer = true, outCovValid = false; } // every mutating member
    // Deferred propagation: VIOFilter::integrateUpToTime (src/VIOFilter.cpp:140-178) calls integrateRiccatiStateFast once and then integrateObserverState once per
    // buffered IMU sample, and reads nothing of the filter in between. The binding records those calls and issues them as ONE eqf_propagate_fast (one launch instead
    // of 2 + k) at the first member call that is not an observer step - same arithmetic (eqf_propagate_fast IS that sequence), same results bit for bit.
    bool pendingRiccati = false, pendingDiscrete = false;
    double pendingMean[13] = {0}, pendingDt = 0.0, pendingQd[12] = {0}, pendingPd8[8] = {0};
    std::vector<double> pendingImu, pendingDts; // 13 doubles per deferred observer step
    DeviceTwin() = default;
    DeviceTwin(const DeviceTwin& o);
    DeviceTwin& operator=(const DeviceTwin& o);
    DeviceTwin(DeviceTwin&& o) noexcept;
    DeviceTwin& operator=(DeviceTwin&& o) noexcept;
    ~DeviceTwin();
};
} // namespace eqvio_mi355x

struct VIO_eqf {
    EqFCoordinateSuite const* coordinateSuite = &EqFCoordinateSuite_euclid;
    VIOState xi0;
    VIOGroup X = VIOGroup::Identity();
    Eigen::MatrixXd Sigma = Eigen::MatrixXd::Identity(VIOSensorState::CompDim, VIOSensorState::CompDim);
    double currentTime = -1;

    void addNewLandmarks(std::vector<Landmark>& newLandmarks, const Eigen::MatrixXd& newLandmarkCov);
    void removeLandmarkByIndex(const int& idx);
    void removeLandmarkById(const int& id);
    void removeInvalidLandmarks();
    Eigen::Matrix3d getLandmarkCovById(const int& id) const;
    Eigen::Matrix2d getOutputCovById(const int& id, const Eigen::Vector2d& y, const GIFT::GICameraPtr& camPtr) const;
    void integrateObserverState(const IMUVelocity& imuVelocity, const double& dt, const bool& discreteLift = true);
    void integrateRiccatiStateFast(const IMUVelocity& imuVelocity, const double& dt, const Eigen::Matrix<double, IMUVelocity::CompDim, IMUVelocity::CompDim>& inputGainMatrix,
                                   const Eigen::MatrixXd& stateGainMatrix);
    void integrateRiccatiStateAccurate(const IMUVelocity& imuVelocity, const double& dt, const Eigen::Matrix<double, IMUVelocity::CompDim, IMUVelocity::CompDim>& inputGainMatrix,
                                       const Eigen::MatrixXd& stateGainMatrix);
    void integrateRiccatiStateDiscrete(const IMUVelocity& imuVelocity, const double& dt, const Eigen::Matrix<double, IMUVelocity::CompDim, IMUVelocity::CompDim>& inputGainMatrix,
                                       const Eigen::MatrixXd& stateGainMatrix);
    void performVisionUpdate(const VisionMeasurement& measurement, const Eigen::MatrixXd& outputGainMatrix, const bool& useEquivariantOutput = true,
                             const bool& discreteCorrection = false);
    VIOState stateEstimate() const;
    double computeNEES(const VIOState& trueState) const;

    // ---- ADDED for the MI355X binding (2 of 2). A trailing member with a default initialiser keeps VIO_eqf an aggregate:
    // `VIO_eqf{suite, xi0, X, Sigma}` (test/test_FilterStatistics.cpp:40) and copies still work.
    void pull() const;      // device -> host members, if the device is ahead (VIOFilter::viewEqFState() calls it before returning)
    void markHostEdited() { twin.hostEdited = true; } // after code that assigns xi0 / X / Sigma directly (VIOFilter.cpp:33-108)
    // ---- ADDED for the MI355X binding (3 of 3, optional): the fused entry points of include/eqf_hip.h. A VIOFilter.cpp that calls them
    // (tests/integration/VIOFilter_mi355x_hunks.hpp) takes ONE host wait per frame; without them every member above still works.
    // The gain matrices of VIOFilterSettings.h:176-206 are diagonal: the fused members take their distinct values (12 input gains; the 7 sensor 3-blocks + the
    // per-landmark value of the state gain; the pixel variance) instead of dense (21 + 3N)^2 / (2M)^2 matrices built per frame (3 MB + 1.3 MB at N = 200).
    void propagateFast(const IMUVelocity& meanVelocity, const double& dtTotal, const double (&inputGainDiag)[12], const double (&stateGainDiag8)[8],
                       const std::vector<IMUVelocity>& samples, const std::vector<double>& dts, const bool& discreteLift); // eqf_propagate_fast
    void stageMeasurement(const VisionMeasurement& measurement);                                                                                    // eqf_stage_measurement
    // eqf_stats_then_update (maxOutliers < 0) / eqf_stats_select_update: returns 1 updated, 0 statistics only (absErr / probErr by state index), -1 not applicable
    int statsThenUpdate(const VisionMeasurement& measurement, const double& thrAbs, const double& thrProb, const long& maxOutliers, const double& outputGainVariance,
                        const bool& useEquivariantOutput, const bool& discreteCorrection, std::vector<double>& absErr, std::vector<double>& probErr);
    mutable eqvio_mi355x::DeviceTwin twin;
};
