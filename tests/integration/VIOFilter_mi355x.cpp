// src/VIOFilter_mi355x.cpp — the reference-side binding of INTEGRATION.md §A for `class VIOFilter`: what the hot-path members of src/VIOFilter.cpp
// (processIMUData :58-63, integrateUpToTime :134-192, processVisionData :194-241, addNewLandmarks :258-278, removeOldLandmarks :280-302, removeOutliers
// :304-364, getMedianSceneDepth :366-380) become in a tree whose VIO_eqf is bound to the MI355X (VIO_eqf_mi355x.cpp). Two forms, chosen by the one added
// setting `mi355xFused`:
//   false: the reference's call sequence, MEMBER FOR MEMBER - integrateRiccatiStateFast + k x integrateObserverState, removeOutliers with one
//          getOutputCovById per measured landmark, performVisionUpdate. Nothing but VIO_eqf_mi355x.cpp and the hooks of INTEGRATION.md §A is needed for it.
//   true:  the fused entry points - stageMeasurement, propagateFast, statsThenUpdate (eqf_stage_measurement, eqf_propagate_fast, eqf_stats_then_update /
//          eqf_stats_select_update of include/eqf_hip.h): the whole frame is two C-ABI calls and one host wait. The hunks against src/VIOFilter.cpp are
//          marked FUSED below; everything else is the control flow of the reference, restated over the same member names.
// Written against the common subset of Eigen 3.4 / LiePP / GIFT and of tests/integration/standin/, compiled with -Wall -Wextra -Werror and run by
// tests/test_integration_filter.py and bench.py (tests/integration/run_filter_frames.cpp is the caller, shaped like src/main_sim.cpp:128-184).
#include "eqvio/VIOFilter.h"

#include <algorithm>
#include <cassert>
#include <cmath>
#include <map>
#include <numeric>

LoopTimer loopTimer;

VIOFilter::VIOFilter(const VIOState& xi0, const VIOFilter::Settings& settings, const double& time) { // :43-56
    this->settings = std::make_unique<VIOFilter::Settings>(settings);
    filterState.Sigma = this->settings->constructInitialStateCovariance(xi0.cameraLandmarks.size());
    filterState.xi0 = xi0;
    for (const Landmark& lm : xi0.cameraLandmarks) {
        filterState.X.Q.emplace_back(liepp::SOT3d::Identity());
        filterState.X.id.emplace_back(lm.id);
    }
    filterState.coordinateSuite = getCoordinates(settings.coordinateChoice);
    filterState.currentTime = time;
    filterState.markHostEdited(); // hook of INTEGRATION.md §A: xi0 / X / Sigma were assigned directly
    initialisedFlag = true;
}

void VIOFilter::processIMUData(const IMUVelocity& imuVelocity) { // :58-63 (initialiseFromIMUData is host-only code and stays as it is)
    assert(initialisedFlag);
    velocityBuffer.emplace_back(imuVelocity);
}

bool VIOFilter::integrateUpToTime(const double& newTime) { // :134-192
    if (newTime <= filterState.currentTime || filterState.currentTime < 0 || velocityBuffer.empty())
        return false;
    const auto clipped = [&](size_t i) { // the interval of sample i inside (currentTime, newTime]
        const double t0 = std::max(velocityBuffer.at(i).stamp, filterState.currentTime);
        const double t1 = i + 1 < velocityBuffer.size() ? std::min(velocityBuffer.at(i + 1).stamp, newTime) : newTime;
        return std::max(t1 - t0, 0.0);
    };
    double accumulatedTime = 0;
    IMUVelocity accumulatedVelocity = IMUVelocity::Zero();
    if (settings->fastRiccati) {
        for (size_t i = 0; i < velocityBuffer.size(); ++i) {
            const double dt = clipped(i);
            accumulatedTime += dt;
            accumulatedVelocity = accumulatedVelocity + velocityBuffer.at(i) * dt;
        }
        accumulatedVelocity = accumulatedVelocity * (1.0 / accumulatedTime);
    }
    if (settings->fastRiccati && settings->mi355xFused) {
        // FUSED (replaces :155-158 and the loop :160-178): one call = integrateRiccatiStateFast at the current X followed by every integrateObserverState
        std::vector<double> dts(velocityBuffer.size());
        for (size_t i = 0; i < velocityBuffer.size(); ++i)
            dts[i] = clipped(i);
        // (the distinct diagonal values of constructInputGainMatrix / constructStateGainMatrix, VIOFilterSettings.h:176-201, instead of the dense matrices)
        const double Qd[12] = {settings->velGyrNoise * settings->velGyrNoise, settings->velGyrNoise * settings->velGyrNoise, settings->velGyrNoise * settings->velGyrNoise,
                               settings->velAccNoise * settings->velAccNoise, settings->velAccNoise * settings->velAccNoise, settings->velAccNoise * settings->velAccNoise,
                               settings->velGyrBiasWalk * settings->velGyrBiasWalk, settings->velGyrBiasWalk * settings->velGyrBiasWalk, settings->velGyrBiasWalk * settings->velGyrBiasWalk,
                               settings->velAccBiasWalk * settings->velAccBiasWalk, settings->velAccBiasWalk * settings->velAccBiasWalk, settings->velAccBiasWalk * settings->velAccBiasWalk};
        const double Pd8[8] = {settings->biasOmegaProcessVariance, settings->biasAccelProcessVariance, settings->attitudeProcessVariance, settings->positionProcessVariance,
                               settings->velocityProcessVariance, settings->cameraAttitudeProcessVariance, settings->cameraPositionProcessVariance, settings->pointProcessVariance};
        filterState.propagateFast(accumulatedVelocity, accumulatedTime, Qd, Pd8, velocityBuffer, dts, settings->useDiscreteVelocityLift);
    } else {
        if (settings->fastRiccati)
            filterState.integrateRiccatiStateFast(accumulatedVelocity, accumulatedTime, settings->constructInputGainMatrix(),
                                                  settings->constructStateGainMatrix(filterState.xi0.cameraLandmarks.size()));
        for (size_t i = 0; i < velocityBuffer.size(); ++i) {
            const double dt = clipped(i);
            if (!settings->fastRiccati && dt > 0) {
                if (settings->useDiscreteStateMatrix)
                    filterState.integrateRiccatiStateDiscrete(velocityBuffer.at(i), dt, settings->constructInputGainMatrix(),
                                                              settings->constructStateGainMatrix(filterState.xi0.cameraLandmarks.size()));
                else
                    filterState.integrateRiccatiStateAccurate(velocityBuffer.at(i), dt, settings->constructInputGainMatrix(),
                                                              settings->constructStateGainMatrix(filterState.xi0.cameraLandmarks.size()));
            }
            filterState.integrateObserverState(velocityBuffer.at(i), dt, settings->useDiscreteVelocityLift);
        }
    }
    filterState.currentTime = newTime;
    auto it = std::find_if(velocityBuffer.begin(), velocityBuffer.end(), [this](const IMUVelocity& imuVel) { return imuVel.stamp >= this->filterState.currentTime; });
    if (it != velocityBuffer.begin()) {
        --it;
        velocityBuffer.erase(velocityBuffer.begin(), it);
    }
    return true;
}

void VIOFilter::processVisionData(const VisionMeasurement& measurement) { // :194-241
    loopTimer.startTiming("propagation");
    if (settings->mi355xFused && initialisedFlag && settings->fastRiccati)
        filterState.stageMeasurement(measurement); // FUSED (new first line): the measurement travels to HBM inside the propagation kernel
    const bool integrationFlag = integrateUpToTime(measurement.stamp);
    if (!integrationFlag || !initialisedFlag)
        return;
    loopTimer.endTiming("propagation");

    loopTimer.startTiming("preprocessing");
    if (settings->removeLostLandmarks)
        removeOldLandmarks(measurement.getIds());
    if (settings->mi355xFused) {
        // FUSED (replaces :213-233 when it applies): with a fixed initial depth a new landmark depends on its pixel only and the outlier test never looks at
        // it, so the new landmarks can be appended BEFORE the test; the outlier statistics, (where needed) the outlier decision and the update are then queued
        // back to back with one host wait. r == 0: the device found outlier candidates and left the decision to the code below (statistics in hand);
        // r == -1: not applicable. Same state as the reference's order either way (tests/test_integration_filter.py, against the oracle).
        const bool earlyAdd = !settings->useMedianDepth;
        if (earlyAdd)
            addNewLandmarks(measurement);
        if (!measurement.camCoordinates.empty() && !filterState.X.id.empty()) {
            const long maxOutliers = earlyAdd ? (long)(size_t)((1.0 - settings->featureRetention) * measurement.camCoordinates.size()) : -1;
            std::vector<double> absErr, probErr;
            const int r = filterState.statsThenUpdate(measurement, settings->outlierThresholdAbs, settings->outlierThresholdProb, maxOutliers,
                                                      settings->measurementNoise * settings->measurementNoise, settings->useEquivariantOutput,
                                                      settings->useDiscreteInnovationLift, absErr, probErr);
            if (r == 1) {
                loopTimer.endTiming("preprocessing");
                loopTimer.startTiming("correction");
                filterState.removeInvalidLandmarks();
                loopTimer.endTiming("correction");
                return;
            }
        }
        VisionMeasurement matchedMeasurement = measurement;
        removeOutliers(matchedMeasurement);
        if (!earlyAdd)
            addNewLandmarks(matchedMeasurement);
        loopTimer.endTiming("preprocessing");
        if (matchedMeasurement.camCoordinates.empty())
            return;
        loopTimer.startTiming("correction");
        filterState.performVisionUpdate(matchedMeasurement, settings->constructOutputGainMatrix(matchedMeasurement.camCoordinates.size()), settings->useEquivariantOutput,
                                        settings->useDiscreteInnovationLift);
        filterState.removeInvalidLandmarks();
        loopTimer.endTiming("correction");
        return;
    }
    VisionMeasurement matchedMeasurement = measurement;
    removeOutliers(matchedMeasurement);
    addNewLandmarks(matchedMeasurement);
    loopTimer.endTiming("preprocessing");
    if (matchedMeasurement.camCoordinates.empty())
        return;
    loopTimer.startTiming("correction");
    filterState.performVisionUpdate(matchedMeasurement, settings->constructOutputGainMatrix(matchedMeasurement.camCoordinates.size()), settings->useEquivariantOutput,
                                    settings->useDiscreteInnovationLift);
    filterState.removeInvalidLandmarks();
    loopTimer.endTiming("correction");
}

VIOState VIOFilter::stateEstimate() const { return filterState.stateEstimate(); }
const VIO_eqf& VIOFilter::viewEqFState() const {
    filterState.pull(); // hook of INTEGRATION.md §A: the writers read Sigma / X / xi0 through this reference
    return filterState;
}
double VIOFilter::getTime() const { return filterState.currentTime; }

void VIOFilter::addNewLandmarks(const VisionMeasurement& measurement) { // :258-278
    std::vector<Landmark> newLandmarks;
    std::vector<int> have = filterState.X.id; // sorted copy: O(M log N) membership (the reference scans X.id per feature: 40 000 comparisons per frame at N = 200)
    std::sort(have.begin(), have.end());
    for (const auto& cc : measurement.camCoordinates) {
        const int ccId = cc.first;
        if (!std::binary_search(have.begin(), have.end(), ccId)) {
            Landmark lm;
            lm.p = measurement.cameraPtr->undistortPoint(cc.second);
            lm.id = ccId;
            newLandmarks.emplace_back(lm);
        }
    }
    if (newLandmarks.empty())
        return;
    const double initialDepth = settings->useMedianDepth ? getMedianSceneDepth() : settings->initialSceneDepth;
    for (Landmark& blm : newLandmarks)
        blm.p *= initialDepth;
    const int newN = (int)newLandmarks.size();
    const Eigen::MatrixXd newLandmarksCov = Eigen::MatrixXd::Identity(3 * newN, 3 * newN) * settings->initialPointVariance;
    filterState.addNewLandmarks(newLandmarks, newLandmarksCov);
}

void VIOFilter::removeOldLandmarks(const std::vector<int>& measurementIds) { // :280-302
    std::vector<int> lostIndices; // measurementIds come from a std::map: ascending, so membership is a binary search
    for (int i = 0; i < (int)filterState.X.id.size(); ++i)
        if (!std::binary_search(measurementIds.begin(), measurementIds.end(), filterState.X.id[i]))
            lostIndices.push_back(i);
    for (auto li = lostIndices.rbegin(); li != lostIndices.rend(); ++li) // descending order
        filterState.removeLandmarkByIndex(*li);
}

void VIOFilter::removeOutliers(VisionMeasurement& measurement) { // :304-364
    const size_t maxOutliers = (1.0 - settings->featureRetention) * measurement.camCoordinates.size();
    const VIOState xiHat = stateEstimate();
    const VisionMeasurement yHat = measureSystemState(xiHat, measurement.cameraPtr);
    std::vector<int> proposedOutliers;
    std::map<int, double> absoluteOutliers, probabilisticOutliers;
    for (const auto& [lmId, yHat_i] : yHat.camCoordinates) {
        if (measurement.camCoordinates.count(lmId) == 0)
            continue;
        const double bearingErrorAbs = (measurement.camCoordinates.at(lmId) - yHat_i).norm();
        if (bearingErrorAbs > settings->outlierThresholdAbs) {
            absoluteOutliers[lmId] = bearingErrorAbs;
            proposedOutliers.emplace_back(lmId);
        }
    }
    const VisionMeasurement measurementResidual = measurement - yHat;
    for (const auto& [lmId, yTilde_i] : measurementResidual.camCoordinates) {
        if (absoluteOutliers.count(lmId))
            continue;
        const Eigen::Matrix2d outputCov = filterState.getOutputCovById(lmId, measurement.camCoordinates[lmId], measurement.cameraPtr); // one device round trip each
        const Eigen::Vector2d weighted = outputCov.inverse() * yTilde_i;
        const double bearingErrorProb = yTilde_i.dot(weighted);
        if (bearingErrorProb > settings->outlierThresholdProb) {
            probabilisticOutliers[lmId] = bearingErrorProb;
            proposedOutliers.emplace_back(lmId);
        }
    }
    // absolute outliers first (largest error first), then probabilistic ones: the reference's comparator + reverse (:339-357)
    std::sort(proposedOutliers.begin(), proposedOutliers.end(), [&](const int& lmId1, const int& lmId2) {
        const bool a1 = absoluteOutliers.count(lmId1) > 0, a2 = absoluteOutliers.count(lmId2) > 0;
        if (a1 != a2)
            return a2;
        return a1 ? absoluteOutliers.at(lmId1) < absoluteOutliers.at(lmId2) : probabilisticOutliers.at(lmId1) < probabilisticOutliers.at(lmId2);
    });
    std::reverse(proposedOutliers.begin(), proposedOutliers.end());
    if (proposedOutliers.size() > maxOutliers)
        proposedOutliers.erase(proposedOutliers.begin() + maxOutliers, proposedOutliers.end());
    for (const int& lmId : proposedOutliers) {
        filterState.removeLandmarkById(lmId);
        measurement.camCoordinates.erase(lmId);
    }
}

double VIOFilter::getMedianSceneDepth() const { // :366-380
    const std::vector<Landmark> landmarks = this->stateEstimate().cameraLandmarks;
    std::vector<double> depthsSquared(landmarks.size());
    std::transform(landmarks.begin(), landmarks.end(), depthsSquared.begin(), [](const Landmark& blm) { return blm.p.squaredNorm(); });
    const auto midway = depthsSquared.begin() + depthsSquared.size() / 2;
    std::nth_element(depthsSquared.begin(), midway, depthsSquared.end());
    return midway == depthsSquared.end() ? settings->initialSceneDepth : std::pow(*midway, 0.5);
}
