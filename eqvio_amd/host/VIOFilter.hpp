// Host-side mirror of the reference's filter interface for the MI355X EqF path.
//
// Same class names, member names, argument meaning and error behaviour as the reference
//   class VIOFilter              include/eqvio/VIOFilter.h:36-192      (src/VIOFilter.cpp)
//   struct VIOFilter::Settings   include/eqvio/VIOFilterSettings.h:58-124
//   struct VIO_eqf               include/eqvio/mathematical/VIO_eqf.h:34-134
//   struct VIOState / VIOSensorState / Landmark   include/eqvio/mathematical/VIOState.h:41-90
//   struct VIOGroup              include/eqvio/mathematical/VIOGroup.h:32-70
//   struct IMUVelocity           include/eqvio/mathematical/IMUVelocity.h:33-84
//   struct VisionMeasurement     include/eqvio/mathematical/VisionMeasurement.h:35-62
//   class LoopTimer, loopTimer   include/eqvio/LoopTimer.h:34-95
// but without Eigen / LiePP / GIFT types (absent from this image): vectors are eqf::V3, rotations eqf::Qt,
// poses eqf::Pose. All O(n^2)/O(n^3) arithmetic is delegated to the device through the C-ABI of
// include/eqf_hip.h; this file holds only the reference's control logic (IMU buffering, landmark
// bookkeeping, outlier policy). Everything lives in namespace eqvio_amd.
#pragma once
#include "../csrc/eqf_math.hpp"
#include "eqf_hip.h"
#include <array>
#include <chrono>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace eqvio_amd {

using eqf::Pose;
using eqf::Qt;
using eqf::V3;

// ---- LoopTimer (include/eqvio/LoopTimer.h, src/LoopTimer.cpp). startTiming on an uninitialised label throws
// std::out_of_range exactly like the reference's map::at (LoopTimer.cpp:20).
class LoopTimer {
  public:
    using timer_clock = std::chrono::steady_clock;
    using timer_duration = std::chrono::duration<double>;
    struct LoopTimingData {
        timer_duration loopTimeStart{};
        std::map<std::string, timer_duration> timings;
    };
    void startLoop();
    void startTiming(const std::string& label);
    void endTiming(const std::string& label);
    void initialise(const std::vector<std::string>& headers);
    const LoopTimingData& getLoopTimingData() const { return currentLoopTimingData; }

  protected:
    const timer_clock::time_point timerOrigin = timer_clock::now();
    std::map<std::string, timer_clock::time_point> timerStartPoints;
    LoopTimingData currentLoopTimingData;
};
// The reference has ONE process-global timer (LoopTimer.h:95); here it is thread_local so several filters
// (one per GPU, one thread each) can live in one process (SURVEY.md §8b "Threading").
extern thread_local LoopTimer loopTimer;

struct Landmark {
    V3 p{0, 0, 0};
    int id = -1;
};
struct VIOSensorState {
    std::array<double, 6> inputBias{};
    Pose pose = eqf::pose_identity();
    V3 velocity{0, 0, 0};
    Pose cameraOffset = eqf::pose_identity();
    V3 gravityDir() const { return eqf::q_rot(eqf::q_inv(pose.R), V3{0, 0, 1}); }
    static constexpr int CompDim = 21;
};
struct VIOState {
    VIOSensorState sensor;
    std::vector<Landmark> cameraLandmarks;
    std::vector<int> getIds() const;
    int Dim() const { return VIOSensorState::CompDim + 3 * (int)cameraLandmarks.size(); }
};
struct SOT3 {
    Qt R = eqf::q_identity();
    double a = 1.0;
};
struct VIOGroup {
    std::array<double, 6> beta{};
    Pose A = eqf::pose_identity();
    V3 w{0, 0, 0};
    Pose B = eqf::pose_identity();
    std::vector<SOT3> Q;
    std::vector<int> id;
};
struct IMUVelocity {
    double stamp = 0;
    V3 gyr{0, 0, 0}, acc{0, 0, 0}, gyrBiasVel{0, 0, 0}, accBiasVel{0, 0, 0};
    static IMUVelocity Zero() { return IMUVelocity(); }
    IMUVelocity operator+(const IMUVelocity& other) const;
    IMUVelocity operator*(const double& c) const;
    void pack(double* v13) const;
};
// GIFT::GICamera contract on the three models of include/eqvio_types.h (pinhole, radial-tangential, equidistant)
struct Camera {
    eqvio_camera c{};
    eqf::Cam model() const {
        eqf::Cam k{c.fx, c.fy, c.cx, c.cy};
        k.model = c.model;
        for (int i = 0; i < 5; ++i)
            k.d[i] = c.dist[i];
        return k;
    }
    void projectPoint(V3 p, double& u, double& v) const { eqf::cam_project(model(), p, u, v); }
    V3 undistortPoint(double u, double v) const { return eqf::cam_undistort(model(), u, v); }
};
using GICameraPtr = std::shared_ptr<const Camera>;
// VIOState.cpp:28-68: one step of the VIO dynamics on a state (no covariance), used by predictState
VIOState integrateSystemFunction(const VIOState& state, const IMUVelocity& velocity, const double& dt);
struct VisionMeasurement {
    double stamp = 0;
    std::map<int, std::array<double, 2>> camCoordinates; // ascending id == the reference's row order
    GICameraPtr cameraPtr;
    std::vector<int> getIds() const;
    // The same measurement as two flat arrays (ids ascending, pixels u0 v0 u1 v1 ...), the form the C-ABI takes. Every access validates the cached
    // arrays against the map (all ids, all pixel values: the map is public and may have been edited) and rebuilds them on the first difference;
    // flat() hands out both arrays after ONE such walk. The replay path builds the cache once per frame, before the timed region (eqvio_frames_create).
    std::pair<const std::vector<int>*, const std::vector<double>*> flat() const { return refreshFlat(), std::make_pair(&flatIds_, &flatY_); }
    const std::vector<int>& flatIds() const { return refreshFlat(), flatIds_; }
    const std::vector<double>& flatY() const { return refreshFlat(), flatY_; }
    void invalidateFlat() const { flatIds_.clear(), flatY_.clear(), flatN_ = (size_t)-1; }
    size_t flatRebuilds() const { return flatRebuilds_; } // how often a walk found the cache different from the map
    // The cached arrays WITHOUT the validating walk (built if there are none of the right size): for a consumer that treats them as a hint and checks them
    // against validated data later - eqf_stage_measurement: the staged copy is compared with the measurement of the update call (eqf_stats_then_update) and
    // ignored if it differs. Takes the 1.7 us walk over the std::map off the host path between the doorbell and the propagation's launch.
    std::pair<const std::vector<int>*, const std::vector<double>*> flatHint() const {
        if (flatN_ != camCoordinates.size() || flatIds_.size() != camCoordinates.size())
            refreshFlat();
        return std::make_pair(&flatIds_, &flatY_);
    }

    // One validating walk for the duration of a call that holds the measurement by const reference (VIOFilter::processVisionData asks for the flat arrays three
    // times per frame - lost landmarks, new landmarks, the update: 1.7 us per walk over the std::map). Nobody can edit the map while the guard's owner runs.
    struct Validated {
        const VisionMeasurement& m;
        explicit Validated(const VisionMeasurement& meas) : m(meas) {
            m.refreshFlat();
            m.flatTrusted_.on = true;
        }
        ~Validated() { m.flatTrusted_.on = false; }
        Validated(const Validated&) = delete;
        Validated& operator=(const Validated&) = delete;
    };

  private:
    void refreshFlat() const;
    // The trust a Validated guard lends is the guarded OBJECT's for the guard's lifetime: a copy made meanwhile (processVisionData's matchedMeasurement) starts untrusted -
    // nothing would reset a copied flag, and its refreshFlat() would skip the validating walk for good
    struct Trust {
        bool on = false;
        Trust() = default;
        Trust(const Trust&) {}
        Trust& operator=(const Trust&) {
            on = false;
            return *this;
        }
    };
    mutable Trust flatTrusted_;
    mutable std::vector<int> flatIds_;
    mutable std::vector<double> flatY_;
    mutable size_t flatN_ = (size_t)-1;
    mutable size_t flatRebuilds_ = 0;
};

enum class CoordinateChoice { Euclidean = 0, InvDepth = 1, Normal = 2 };

// Dense column-major matrix owner for the public Sigma view (Eigen::MatrixXd stand-in: rows/cols/operator()/block)
struct MatrixXd {
    int r = 0, c = 0;
    std::vector<double> d;
    int rows() const { return r; }
    int cols() const { return c; }
    double& operator()(int i, int j) { return d[(size_t)j * r + i]; }
    const double& operator()(int i, int j) const { return d[(size_t)j * r + i]; }
};

// ---- VIO_eqf on the device. The reference's public data members xi0 / X / Sigma (VIO_eqf.h:36-42) become
// accessors that read the device-resident state back on demand.
struct VIO_eqf {
    eqf_ctx* ctx = nullptr;
    CoordinateChoice coordinateChoice = CoordinateChoice::Euclidean;
    double currentTime = -1;

    VIO_eqf() = default;
    VIO_eqf(const VIO_eqf&) = delete;
    VIO_eqf& operator=(const VIO_eqf&) = delete;
    ~VIO_eqf();
    void create(int device, int maxLandmarks, CoordinateChoice cc);

    // state views
    VIOState xi0() const;
    VIOGroup X() const;
    MatrixXd Sigma() const;
    const std::vector<int>& ids() const { return ids_; }
    int numLandmarks() const { return (int)ids_.size(); }
    void set(const VIOState& xi0, const VIOGroup& X);
    void setSigma(const MatrixXd& S);
    void setSigmaDiag(const std::vector<double>& diag);

    // VIO_eqf members (include/eqvio/mathematical/VIO_eqf.h:44-134)
    bool addNewLandmarks(std::vector<Landmark>& newLandmarks, double newLandmarkVar, bool held = false); // held: eqf_add_landmarks_held; false: refused, nothing added
    bool holdSupported() const { return eqf_hold_supported(ctx) == 1; }
    void removeLandmarkByIndex(const int& idx);
    void removeLandmarkById(const int& id);
    void removeLandmarksByIndex(const std::vector<int>& idx); // batched form of the above
    void removeInvalidLandmarks();
    int settleInvalid() const; // a removal of invalid landmarks deferred past an unsettled update (EQF_OPT_EARLY_DOORBELL) happens now; returns how many left
    bool removeUnmeasured(const std::vector<int>& measurementIds); // false: ids not ascending, nothing done
    bool sameAsMapped(const std::vector<int>& measurementIds) const; // exactly the ids the last update mapped, one per landmark
    bool findUnknownIds(const std::vector<int>& measurementIds, std::vector<int>& unknownJ, int& n) const; // false: ids not ascending
    std::array<double, 9> getLandmarkCovById(const int& id) const;
    void integrateObserverState(const IMUVelocity& imuVelocity, const double& dt, const bool& discreteLift = true);
    void integrateObserverStates(const std::vector<IMUVelocity>& imus, const std::vector<double>& dts, bool discreteLift); // batched
    void integrateRiccatiStateFast(const IMUVelocity& imuVelocity, const double& dt, const std::array<double, 12>& inputGainDiag, const std::array<double, 8>& stateGainDiag8);
    // integrateRiccatiStateFast(mean sample) followed by integrateObserverStates(all samples) in one device call (eqf_propagate_fast)
    void propagateFast(const IMUVelocity& meanVelocity, const double& dtTotal, const std::array<double, 12>& inputGainDiag, const std::array<double, 8>& stateGainDiag8,
                       const std::vector<IMUVelocity>& imus, const std::vector<double>& dts, bool discreteLift);
    void integrateRiccatiStateAccurate(const IMUVelocity& imuVelocity, const double& dt, const std::array<double, 12>& inputGainDiag, const std::array<double, 8>& stateGainDiag8);
    void integrateRiccatiStateDiscrete(const IMUVelocity& imuVelocity, const double& dt, const std::array<double, 12>& inputGainDiag, const std::array<double, 8>& stateGainDiag8);
    void performVisionUpdate(const VisionMeasurement& measurement, double outputGainVar, const bool& useEquivariantOutput = true, const bool& discreteCorrection = false);
    VIOState stateEstimate() const;
    VIOState predictState(const double& stamp, const std::vector<IMUVelocity>& imuVelocities) const; // VIO_eqf.cpp:139-151 (host: O(kN))
    double computeNEES(const VIOState& trueState) const; // VIO_eqf.cpp:153-170, factorised on the device
    // per-landmark quantities VIOFilter::removeOutliers / getMedianSceneDepth need, all landmarks at once
    void outlierStats(const VisionMeasurement& measurement, std::vector<double>& absErr, std::vector<double>& probErr, std::vector<double>& depth2) const;
    // eqf_stats_then_update: the statistics and (unless an outlier candidate cancels it on the device) the update, one host wait.
    // Returns 1 when the update was performed, 0 when the device cancelled it (statistics valid), -1 when not applicable
    // (a measurement id is not in the state; nothing computed).
    void stageMeasurement(const VisionMeasurement& measurement); // eqf_stage_measurement: hint ahead of the propagation of the same frame
    int statsThenUpdate(const VisionMeasurement& measurement, double thrAbs, double thrProb, double outputGainVar, bool useEquivariantOutput, bool discreteCorrection,
                         std::vector<double>& absErr, std::vector<double>& probErr, std::vector<double>& depth2, long maxOutliers = -1);

  private:
    mutable std::vector<int> ids_;
    mutable bool invalidPending_ = false;
    std::vector<int> scratchIdx_;
    void check(int rc, const char* what) const;
};

class VIOFilter {
  protected:
    VIO_eqf filterState;
    bool initialisedFlag = false;
    std::vector<IMUVelocity> velocityBuffer;

    bool integrateUpToTime(const double& newTime);
    bool addNewLandmarks(const VisionMeasurement& measurement, const std::vector<double>* depth2, bool held = false);
    void removeOldLandmarks(const std::vector<int>& measurementIds);
    void removeOutliers(VisionMeasurement& measurement, std::vector<double>& depth2, const std::vector<double>* absErrIn = nullptr,
                        const std::vector<double>* probErrIn = nullptr);
    double getMedianSceneDepth(const std::vector<double>* depth2) const;
    std::vector<int> unknownScratch_; // addNewLandmarks: positions of the measured ids without a landmark

  public:
    struct Settings;
    std::unique_ptr<VIOFilter::Settings> settings;

    VIOFilter() = default;
    explicit VIOFilter(const VIOFilter::Settings& settings);
    VIOFilter(const VIOState& xi0, const VIOFilter::Settings& settings, const double& time = 0.0);
    ~VIOFilter();

    void initialiseFromIMUData(const IMUVelocity& imuVelocity);
    void setState(const VIOState& xi);
    void setLandmarks(const std::vector<Landmark>& cameraLandmarks);
    void augmentLandmarkStates(const std::vector<int>& newIds, const VIOState& providedState);
    void processIMUData(const IMUVelocity& imuVelocity);
    void processVisionData(const VisionMeasurement& measurement);
    double getTime() const;
    bool isInitialised() const { return initialisedFlag; }
    VisionMeasurement getFeaturePredictions(const GICameraPtr& camPtr, const double& stamp = -1);
    VIOState stateEstimate() const;
    const VIO_eqf& viewEqFState() const;
    VIO_eqf& eqfState() { return filterState; }
};

// VIOFilter::Settings (include/eqvio/VIOFilterSettings.h:58-124): same fields and defaults; the gain matrices
// are diagonal in the reference (:176-229) so only their diagonals are constructed.
struct VIOFilter::Settings {
    double biasOmegaProcessVariance = 0.001, biasAccelProcessVariance = 0.001, attitudeProcessVariance = 0.001, positionProcessVariance = 0.001,
           velocityProcessVariance = 0.001, cameraAttitudeProcessVariance = 0.001, cameraPositionProcessVariance = 0.001, pointProcessVariance = 0.001;
    double velGyrNoise = 1e-4, velAccNoise = 1e-3, velGyrBiasWalk = 1e-5, velAccBiasWalk = 1e-3;
    double measurementNoise = 2.0, outlierThresholdAbs = 1e8, outlierThresholdProb = 1e8, featureRetention = 0.3;
    double initialAttitudeVariance = 1.0e-4, initialPositionVariance = 1.0e-4, initialVelocityVariance = 1.0e-2, initialCameraAttitudeVariance = 1.0e-5,
           initialCameraPositionVariance = 1.0e-4, initialPointVariance = 1.0, initialPointDepthVariance = -1.0, initialBiasOmegaVariance = 0.1,
           initialBiasAccelVariance = 0.1, initialSceneDepth = 1.0;
    bool useDiscreteInnovationLift = true, useDiscreteVelocityLift = true, useDiscreteStateMatrix = false, fastRiccati = false, useMedianDepth = true,
         useFeaturePredictions = false, useEquivariantOutput = true, removeLostLandmarks = true;
    CoordinateChoice coordinateChoice = CoordinateChoice::Euclidean;
    Pose cameraOffset = eqf::pose_identity();
    // device placement (not in the reference)
    int device = 0;
    int maxLandmarks = 256;

    Settings() = default;
    explicit Settings(const eqvio_settings& s);
    std::vector<double> constructInitialStateCovarianceDiag(const size_t& numLandmarks = 0) const;
    std::array<double, 8> constructStateGainDiag8() const;
    std::array<double, 12> constructInputGainDiag() const;
    double constructOutputGainVar() const { return measurementNoise * measurementNoise; }
};

} // namespace eqvio_amd
