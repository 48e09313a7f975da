// TEST SCAFFOLDING. `VIOFilter::Settings` re-declared with the reference's names, members and signatures (include/eqvio/VIOFilterSettings.h:58-229) as far as
// tests/integration/VIOFilter_mi355x_hunks.hpp and the replay driver touch them, over the stand-in value types of eqvio/mathematical/VIO_eqf.h. The ONE marked field
// is the only addition the fused binding needs.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "eqvio/mathematical/VIO_eqf.h"

enum class CoordinateChoice { Euclidean, InvDepth, Normal };
inline const EqFCoordinateSuite* getCoordinates(const CoordinateChoice& cc) { // EqFMatrices.h:81-90
    return cc == CoordinateChoice::Euclidean ? &EqFCoordinateSuite_euclid : cc == CoordinateChoice::InvDepth ? &EqFCoordinateSuite_invdepth : &EqFCoordinateSuite_normal;
}
class VIOFilter { // only the nested settings type is needed here: the driver of this test replays the member sequence of src/VIOFilter.cpp from a plan
  public:
    struct Settings;
};

struct VIOFilter::Settings { // VIOFilterSettings.h:58-124 (the fields of the eqf block that the hot path reads)
    double biasOmegaProcessVariance = 0.001, biasAccelProcessVariance = 0.001, attitudeProcessVariance = 0.001, positionProcessVariance = 0.001, velocityProcessVariance = 0.001;
    double cameraAttitudeProcessVariance = 0.001, cameraPositionProcessVariance = 0.001, pointProcessVariance = 0.001;
    double velGyrNoise = 1e-4, velAccNoise = 1e-3, velGyrBiasWalk = 1e-5, velAccBiasWalk = 1e-3;
    double measurementNoise = 2.0, outlierThresholdAbs = 1e8, outlierThresholdProb = 1e8, featureRetention = 0.3;
    double initialAttitudeVariance = 1e-4, initialPositionVariance = 1e-4, initialVelocityVariance = 1e-2, initialCameraAttitudeVariance = 1e-5, initialCameraPositionVariance = 1e-4;
    double initialPointVariance = 1.0, initialPointDepthVariance = -1.0, initialBiasOmegaVariance = 0.1, initialBiasAccelVariance = 0.1, initialSceneDepth = 1.0;
    bool useDiscreteInnovationLift = true, useDiscreteVelocityLift = true, useDiscreteStateMatrix = false, fastRiccati = false, useMedianDepth = true;
    bool useFeaturePredictions = false, useEquivariantOutput = true, removeLostLandmarks = true;
    CoordinateChoice coordinateChoice = CoordinateChoice::Euclidean;
    liepp::SE3d cameraOffset = liepp::SE3d::Identity();
    // ---- ADDED for the MI355X binding: false = the reference's call sequence, member for member; true = the fused entry points
    bool mi355xFused = false;

    Eigen::MatrixXd constructInitialStateCovariance(const size_t& numLandmarks = 0) const { // :208-229
        const int n = VIOSensorState::CompDim + 3 * (int)numLandmarks;
        Eigen::MatrixXd S = Eigen::MatrixXd::Zero(n, n);
        const double v[7] = {initialBiasOmegaVariance, initialBiasAccelVariance, initialAttitudeVariance, initialPositionVariance, initialVelocityVariance, initialCameraAttitudeVariance,
                             initialCameraPositionVariance};
        for (int i = 0; i < VIOSensorState::CompDim; ++i) S(i, i) = v[i / 3];
        for (int i = VIOSensorState::CompDim; i < n; ++i) S(i, i) = (initialPointDepthVariance > 0 && (i - VIOSensorState::CompDim) % 3 == 2) ? initialPointDepthVariance : initialPointVariance;
        return S;
    }
    Eigen::MatrixXd constructStateGainMatrix(const size_t& numLandmarks) const { // :176-190
        const int n = VIOSensorState::CompDim + 3 * (int)numLandmarks;
        Eigen::MatrixXd P = Eigen::MatrixXd::Zero(n, n);
        const double v[7] = {biasOmegaProcessVariance, biasAccelProcessVariance, attitudeProcessVariance, positionProcessVariance, velocityProcessVariance, cameraAttitudeProcessVariance,
                             cameraPositionProcessVariance};
        for (int i = 0; i < VIOSensorState::CompDim; ++i) P(i, i) = v[i / 3];
        for (int i = VIOSensorState::CompDim; i < n; ++i) P(i, i) = pointProcessVariance;
        return P;
    }
    Eigen::Matrix<double, 12, 12> constructInputGainMatrix() const { // :192-201
        Eigen::Matrix<double, 12, 12> Q;
        const double v[4] = {velGyrNoise * velGyrNoise, velAccNoise * velAccNoise, velGyrBiasWalk * velGyrBiasWalk, velAccBiasWalk * velAccBiasWalk};
        for (int i = 0; i < 12; ++i) Q(i, i) = v[i / 3];
        return Q;
    }
    Eigen::MatrixXd constructOutputGainMatrix(const size_t& numLandmarks) const { // :203-206
        return Eigen::MatrixXd::Identity(2 * (int)numLandmarks, 2 * (int)numLandmarks) * (measurementNoise * measurementNoise);
    }
};
