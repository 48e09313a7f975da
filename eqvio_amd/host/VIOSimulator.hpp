// Synthetic VIO world for the MI355X EqF path: the caller side of the hot path.
//
// Mirrors the reference's
//   class VIOSimulator           include/eqvio/VIOSimulator.h:29-106         (src/VIOSimulator.cpp)
//   class SimulationDataServer   include/eqvio/dataserver/SimulationDataServer.h:25-70  (src/dataserver/SimulationDataServer.cpp)
// with the same member names and the same measurement model (numerical differentiation of a stamped pose list
// for the IMU, lowest-id visible points for vision), on this repo's value types (no Eigen / LiePP / GIFT / yaml-cpp).
// Differences, all forced by absent third-party pieces:
//   * settings come from a plain struct (SimSettings) instead of a YAML node;
//   * world points come from std::mt19937_64 instead of Eigen's Random()/rand(): the same distribution (uniform on the
//     selected walls of the trajectory box, then shuffled), not the same sample;
//   * GIFT::PinholeCamera::isInDomain is not in /root/reference (un-vendored submodule): here a point is in the
//     domain iff z > 0 and its projection lies inside the image rectangle.
#pragma once
#include "VIOFilter.hpp"
#include <cstdint>
#include <random>

namespace eqvio_amd {

constexpr double GRAVITY_CONSTANT = 9.80665; // include/eqvio/mathematical/VIOState.h:27

struct StampedPose {
    double t = 0;
    Pose pose = eqf::pose_identity();
};

struct SimSettings { // the "sim:" block of the reference's configuration files
    int numPoints = 1000;
    double wallDistance = 2.0;
    uint32_t randomSeed = 0;
    int numWalls = 1;
    size_t maxFeatures = 30;
    bool initialNoise = false, inputNoise = false, outputNoise = false;
    double duration = 100.0;
    std::string trajectory = "wave"; // wave | square | sine | line
    double imuFreq = 200.0, imageFreq = 20.0;
};

class VIOSimulator {
  protected:
    uint32_t randomSeed = 0;
    std::vector<StampedPose> poses;
    std::vector<Landmark> inertialPoints;
    size_t maxFeatures = 30;
    bool initialNoise = false, inputNoise = false, outputNoise = false;
    VIOFilter::Settings filterSettings;
    mutable std::mt19937_64 noiseRng;

    std::vector<Landmark> generateWorldPoints(const int num = 1000, const double distance = 1.0, const int numWalls = 1) const;
    // columns: inertial position, velocity, acceleration at time ct (cubic through four poses)
    void getInertialStates(size_t it, const double& ct, V3& pos, V3& vel, V3& acc) const;
    size_t getTimeIndex(const double& t) const; // first pose with stamp >= t (poses.size() if none)

  public:
    VisionMeasurement getVision(const double& time) const;
    IMUVelocity getIMU(const double& time, const double& samplingFrequency = -1) const;
    const std::vector<StampedPose>& viewPoses() const { return poses; }
    VIOState getFullState(const double& time = -1, const bool& allowNoise = false) const;

    VIOSimulator() = default;
    VIOSimulator(const std::vector<StampedPose>& poses, const GICameraPtr& camPtr, const SimSettings& settings = SimSettings(),
                 const VIOFilter::Settings& filterSettings = VIOFilter::Settings());

    GICameraPtr cameraPtr;
    Pose cameraOffset = eqf::pose_identity();
};

enum class MeasurementType { Image, IMU, None }; // include/eqvio/dataserver/DataServerBase.h

class SimulationDataServer {
  protected:
    VIOSimulator simulator;
    double imageFreq = 20.0, imuFreq = 200.0, maxSimulationTime = 100.0;
    int imuMeasCount = 0, imageMeasCount = 0;
    double nextImageTime() const { return imageMeasCount / imageFreq; }
    double nextIMUTime() const { return imuMeasCount / imuFreq; }

  public:
    MeasurementType nextMeasurementType() const;
    IMUVelocity getIMU() { return getSimIMU(); }
    double nextTime() const;
    VisionMeasurement getSimVision();
    IMUVelocity getSimIMU();
    VIOState getInitialCondition() const { return simulator.getFullState(0.0, true); }
    VIOState getTrueState(const double& stamp, const bool& withNoise = false) const { return simulator.getFullState(stamp, withNoise); }
    std::shared_ptr<Pose> cameraExtrinsics() const { return std::make_shared<Pose>(simulator.cameraOffset); }
    std::vector<StampedPose> generateTrajectory(const std::string& choice) const;
    const VIOSimulator& viewSimulator() const { return simulator; }

    explicit SimulationDataServer(const SimSettings& simSettings = SimSettings(), const VIOFilter::Settings& filterSettings = VIOFilter::Settings());
};

} // namespace eqvio_amd
