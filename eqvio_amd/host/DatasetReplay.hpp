// Dataset ingestion for the MI355X EqF path without images: IMU and ground-truth text files in
// the two dataset layouts the reference reads, plus precomputed feature tracks in the writer's features.csv format
// (VIOWriter.cpp:82-94), served in stamp order like the reference's SimpleDataServer.
//   ASL / EuRoC   src/dataserver/ASLDatasetReader.cpp:22-53, 104-130   comma separated, one header line, stamps in ns
//                 columns: stamp, gyr xyz, acc xyz [, gyrBiasVel xyz, accBiasVel xyz]  (IMUVelocity.cpp:82-89)
//   UZH-FPV       src/dataserver/UZHFPVDatasetReader.cpp:23-58, 117-139  space separated, one header line, a leading
//                 index column, stamps in seconds
//   ground truth  stamp, position xyz, quaternion wxyz (CSVLine.h:213-215); rows whose stamp does not advance by more
//                 than 1e-8 s are dropped (ASLDatasetReader.cpp:120-125)
// No image decoding and no feature tracker: GIFT / OpenCV are outside the hot path and absent from this image.
#pragma once
#include "VIOSimulator.hpp" // StampedPose, MeasurementType
#include <deque>
#include <fstream>

namespace eqvio_amd {

enum class DatasetFormat { ASL, UZHFPV };

// One delimited text line split into fields (CSVLine.h:33-62): fields are trimmed of blanks and a trailing '\r'
std::vector<std::string> splitLine(const std::string& line, char delim);

class TrackReplayServer {
  protected:
    std::ifstream imuFile, featuresFile;
    DatasetFormat format;
    GICameraPtr cameraPtr;
    double cameraLag = 0.0;
    std::unique_ptr<IMUVelocity> nextIMUData;
    std::unique_ptr<VisionMeasurement> nextImageData;
    std::unique_ptr<IMUVelocity> readIMU();
    std::unique_ptr<VisionMeasurement> readFeatures();

  public:
    TrackReplayServer(const std::string& imuFileName, const std::string& featuresFileName, DatasetFormat format, const GICameraPtr& camera,
                      double cameraLag = 0.0);
    MeasurementType nextMeasurementType() const; // SimpleDataServer.cpp:20-30
    double nextTime() const;                     // :51-59
    IMUVelocity getIMU();                        // :38-42
    VisionMeasurement getSimVision();            // the precomputed tracks stand where the reference tracks an image
    GICameraPtr camera() const { return cameraPtr; }
    static std::vector<StampedPose> groundtruth(const std::string& fileName, DatasetFormat format);
};

// The camera file of a dataset (main_opt.cpp:114-147: intrinsics for the measurement's camera, extrinsics into settings.cameraOffset), the subset of YAML the two
// readers consume - yaml-cpp is not in this image, so the few keys are picked out of the text directly:
//   ASL / EuRoC  mav0/cam0/sensor.yaml (ASLDatasetReader.cpp:76-101): resolution [w, h], intrinsics [fu, fv, cu, cv], distortion_coefficients [k1, k2, p1, p2 (, k3)]
//                -> radial-tangential camera; T_BS: data: [16 numbers, row major] = the pose of the camera w.r.t. the IMU, used as it is
//   UZH-FPV      camchain-imucam-*.yaml (UZHFPVDatasetReader.cpp:78-115), under cam0: resolution, intrinsics, distortion_coeffs [4] -> equidistant camera;
//                T_cam_imu: four rows of four = the pose of the IMU w.r.t. the camera, INVERTED for the offset
// Throws std::runtime_error on a missing key or a short list.
void readCameraFile(const std::string& fileName, DatasetFormat format, Camera& camera, Pose& cameraOffset);

} // namespace eqvio_amd
