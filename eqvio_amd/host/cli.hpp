// Flag parsing shared by the two mains (eqvio_sim, eqvio_opt). The reference reads these values from the "eqf:" block of
// a YAML file (VIOFilterSettings.h:126-174); yaml-cpp is not in this image, so every key is a --flag of the same name.
#pragma once
#include "VIOFilter.hpp"
#include <cstdlib>
#include <functional>
#include <string>

namespace eqvio_amd {

// returns true when `a` was a filter-settings flag (and consumed its value through val())
inline bool parseFilterFlag(const std::string& a, const std::function<const char*()>& val, VIOFilter::Settings& fs) {
    struct D { const char* name; double VIOFilter::Settings::*p; };
    static const D doubles[] = {
        {"--biasOmegaProcessVariance", &VIOFilter::Settings::biasOmegaProcessVariance}, {"--biasAccelProcessVariance", &VIOFilter::Settings::biasAccelProcessVariance},
        {"--attitudeProcessVariance", &VIOFilter::Settings::attitudeProcessVariance}, {"--positionProcessVariance", &VIOFilter::Settings::positionProcessVariance},
        {"--velocityProcessVariance", &VIOFilter::Settings::velocityProcessVariance}, {"--cameraAttitudeProcessVariance", &VIOFilter::Settings::cameraAttitudeProcessVariance},
        {"--cameraPositionProcessVariance", &VIOFilter::Settings::cameraPositionProcessVariance}, {"--pointProcessVariance", &VIOFilter::Settings::pointProcessVariance},
        {"--velGyrNoise", &VIOFilter::Settings::velGyrNoise}, {"--velAccNoise", &VIOFilter::Settings::velAccNoise},
        {"--velGyrBiasWalk", &VIOFilter::Settings::velGyrBiasWalk}, {"--velAccBiasWalk", &VIOFilter::Settings::velAccBiasWalk},
        {"--measurementNoise", &VIOFilter::Settings::measurementNoise}, {"--outlierThresholdAbs", &VIOFilter::Settings::outlierThresholdAbs},
        {"--outlierThresholdProb", &VIOFilter::Settings::outlierThresholdProb}, {"--featureRetention", &VIOFilter::Settings::featureRetention},
        {"--initialAttitudeVariance", &VIOFilter::Settings::initialAttitudeVariance}, {"--initialPositionVariance", &VIOFilter::Settings::initialPositionVariance},
        {"--initialVelocityVariance", &VIOFilter::Settings::initialVelocityVariance}, {"--initialCameraAttitudeVariance", &VIOFilter::Settings::initialCameraAttitudeVariance},
        {"--initialCameraPositionVariance", &VIOFilter::Settings::initialCameraPositionVariance}, {"--initialPointVariance", &VIOFilter::Settings::initialPointVariance},
        {"--initialPointDepthVariance", &VIOFilter::Settings::initialPointDepthVariance}, {"--initialBiasOmegaVariance", &VIOFilter::Settings::initialBiasOmegaVariance},
        {"--initialBiasAccelVariance", &VIOFilter::Settings::initialBiasAccelVariance}, {"--initialSceneDepth", &VIOFilter::Settings::initialSceneDepth},
    };
    struct B { const char* name; bool VIOFilter::Settings::*p; };
    static const B bools[] = {
        {"--useDiscreteInnovationLift", &VIOFilter::Settings::useDiscreteInnovationLift}, {"--useDiscreteVelocityLift", &VIOFilter::Settings::useDiscreteVelocityLift},
        {"--useDiscreteStateMatrix", &VIOFilter::Settings::useDiscreteStateMatrix}, {"--fastRiccati", &VIOFilter::Settings::fastRiccati},
        {"--useMedianDepth", &VIOFilter::Settings::useMedianDepth}, {"--useEquivariantOutput", &VIOFilter::Settings::useEquivariantOutput},
        {"--removeLostLandmarks", &VIOFilter::Settings::removeLostLandmarks},
    };
    for (const D& d : doubles)
        if (a == d.name) {
            fs.*(d.p) = std::atof(val());
            return true;
        }
    for (const B& b : bools)
        if (a == b.name) {
            fs.*(b.p) = std::atoi(val()) != 0;
            return true;
        }
    if (a == "--coordinateChoice") { // coordinateSelection, VIOFilterSettings.h:33-46
        const std::string c = val();
        if (c == "Euclidean")
            fs.coordinateChoice = CoordinateChoice::Euclidean;
        else if (c == "InvDepth")
            fs.coordinateChoice = CoordinateChoice::InvDepth;
        else
            throw std::runtime_error("Invalid coordinate choice. Valid choices on the MI355X path are Euclidean, InvDepth.");
        return true;
    }
    if (a == "--device") {
        fs.device = std::atoi(val());
        return true;
    }
    if (a == "--maxLandmarks") {
        fs.maxLandmarks = std::atoi(val());
        return true;
    }
    return false;
}

} // namespace eqvio_amd
