// Output files of a filter run, in the reference's formats so its analysis scripts read them unchanged.
// Mirrors class VIOWriter (include/eqvio/VIOWriter.h:30-101, src/VIOWriter.cpp:22-228) and the CSV element rules of
// include/eqvio/csv/CSVLine.h:84-99, 153-222:
//   every line is "<stamp>, v1, v2, ..." with the stamp at precision 20 and every other value formatted by a default
//   std::ostream (6 significant digits); SE3 is written position first then quaternion (w, x, y, z).
// Files: IMUState.csv camera.csv bias.csv points.csv features.csv timing.csv landmarkError.csv trueState.csv nees.csv
//        poseConsistency.csv cameraConsistency.csv biasConsistency.csv
// The reference writes asynchronously (aofstream); here plain std::ofstream: the device filter is not waiting on I/O.
#pragma once
#include "VIOFilter.hpp"
#include <fstream>

namespace eqvio_amd {

class VIOWriter {
  protected:
    std::string outputDir;
    std::ofstream IMUStateFile, cameraFile, biasFile, pointsFile, landmarkErrorFile, trueStateFile, neesFile, poseConsistencyFile, cameraConsistencyFile,
        biasConsistencyFile, featuresFile, timingFile;

  public:
    explicit VIOWriter(const std::string& outputDir);
    void writeStates(const double& stamp, const VIOState& xi);
    void writeFeatures(const VisionMeasurement& y);
    void writeTiming(const LoopTimer::LoopTimingData& timingData);
    void writeLandmarkError(const double& stamp, const VIOState& trueState, const VIOState& estState);
    void writeConsistency(const double& stamp, const VIOState& trueState, const VIO_eqf& filter);
};

} // namespace eqvio_amd
