// What a maintainer ADDS to src/VIOFilter.cpp in a tree whose VIO_eqf is bound to the MI355X (VIO_eqf_mi355x.cpp) - and nothing else: the two
// fused hunks as functions over the bound VIO_eqf and the reference's own settings struct. The reference's control flow (integrateUpToTime's interval
// clipping, removeOldLandmarks, removeOutliers, addNewLandmarks) is NOT restated here: this repository's restatement of it lives in
// eqvio_amd/host/VIOFilter.cpp (over the C-ABI), and tests/integration/run_filter_frames.cpp replays the reference's member sequence from a plan.
#pragma once
#include <vector>

#include "eqvio/VIOFilter.h"

namespace eqvio_mi355x {
// Hunk 1, inside VIOFilter::integrateUpToTime (src/VIOFilter.cpp:155-178) when settings->fastRiccati && settings->mi355xFused: the mean velocity, the total
// time and the clipped interval of every buffered sample are what that function has computed anyway; one call then stands for integrateRiccatiStateFast
// at the current X followed by every integrateObserverState. The gain matrices of VIOFilterSettings.h:176-201 are diagonal: their distinct values travel.
inline void fusedPropagation(VIO_eqf& filterState, const VIOFilter::Settings& s, const IMUVelocity& meanVelocity, const double& totalTime, const std::vector<IMUVelocity>& samples,
                             const std::vector<double>& clippedDt) {
    double inputGain[12], stateGain[8] = {s.biasOmegaProcessVariance, s.biasAccelProcessVariance, s.attitudeProcessVariance, s.positionProcessVariance,
                                          s.velocityProcessVariance, s.cameraAttitudeProcessVariance, s.cameraPositionProcessVariance, s.pointProcessVariance};
    const double sd[4] = {s.velGyrNoise, s.velAccNoise, s.velGyrBiasWalk, s.velAccBiasWalk};
    for (int i = 0; i < 12; ++i)
        inputGain[i] = sd[i / 3] * sd[i / 3];
    filterState.propagateFast(meanVelocity, totalTime, inputGain, stateGain, samples, clippedDt, s.useDiscreteVelocityLift);
}
// Hunk 2, inside VIOFilter::processVisionData in place of removeOutliers + performVisionUpdate (src/VIOFilter.cpp:213-233): outlier statistics, (with a fixed
// initial depth) the outlier decision, and the update queued back to back with one host wait. Returns what VIO_eqf::statsThenUpdate returns: 1 = the frame is
// done (only removeInvalidLandmarks is left), 0 = the device found outlier candidates and left the decision to the reference's removeOutliers (absErr /
// probErr are in hand), -1 = not applicable; in the last two cases the reference's own lines run as they are.
inline int fusedStatsAndUpdate(VIO_eqf& filterState, const VIOFilter::Settings& s, const VisionMeasurement& measurement, bool newLandmarksAlreadyAdded, std::vector<double>& absErr,
                               std::vector<double>& probErr) {
    if (measurement.camCoordinates.empty() || filterState.X.id.empty())
        return -1;
    const long maxOutliers = newLandmarksAlreadyAdded ? (long)(size_t)((1.0 - s.featureRetention) * measurement.camCoordinates.size()) : -1;
    return filterState.statsThenUpdate(measurement, s.outlierThresholdAbs, s.outlierThresholdProb, maxOutliers, s.measurementNoise * s.measurementNoise, s.useEquivariantOutput,
                                       s.useDiscreteInnovationLift, absErr, probErr);
}
} // namespace eqvio_mi355x
