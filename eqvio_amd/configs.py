"""The reference's shipped filter configurations (configs/*.yaml, eqf block) as Settings objects, and their simulator-consistent stand-ins.
Used by tests/run_configs.py, the parity tests and bench.py --config batch8; no oracle in here."""
from .capi import COORD_EUCLIDEAN, COORD_INVDEPTH, Settings


def template_settings(fast):  # EQVIO_config_template.yaml:1-51 (eqf block)
    s = Settings.defaults()
    for k, v in dict(initialAttitudeVariance=1.0, initialPositionVariance=1.0, initialVelocityVariance=1.0, initialPointVariance=5000.0, initialPointDepthVariance=-1.0,
                     initialCameraAttitudeVariance=0.1, initialCameraPositionVariance=0.1, initialBiasOmegaVariance=1.0, initialBiasAccelVariance=1.0,
                     cameraPositionProcessVariance=1e-4, cameraAttitudeProcessVariance=1e-4, biasOmegaProcessVariance=1e-4, biasAccelProcessVariance=1e-4,
                     attitudeProcessVariance=0.01, positionProcessVariance=0.01, velocityProcessVariance=0.1, pointProcessVariance=0.001, initialSceneDepth=1.0,
                     measurementNoise=0.003, outlierThresholdAbs=0.01, outlierThresholdProb=3.0, featureRetention=0.2,
                     velGyrNoise=1e-4, velAccNoise=1e-4, velGyrBiasWalk=1e-4, velAccBiasWalk=1e-4,
                     fastRiccati=fast, useDiscreteInnovationLift=1, useDiscreteVelocityLift=1, coordinateChoice=COORD_EUCLIDEAN, useMedianDepth=1,
                     useFeaturePredictions=0, useEquivariantOutput=1, removeLostLandmarks=1, useDiscreteStateMatrix=0).items():
        setattr(s, k, v)
    return s


def euroc_settings():  # configs/EQVIO_config_EuRoC_stationary.yaml:17-56
    s = Settings.defaults()
    for k, v in dict(initialSceneDepth=5.00028218320243, initialAttitudeVariance=0.13565029126052572, initialBiasAccelVariance=1.5813333765300104,
                     initialBiasOmegaVariance=97162.79515771076, initialCameraAttitudeVariance=0.0010228558965517584, initialCameraPositionVariance=0.023501400846134893,
                     initialPointVariance=129.90415638150924, initialPositionVariance=0.1, initialVelocityVariance=8.974852995731e-08,
                     measurementNoise=1.9297839969591413, outlierThresholdAbs=4.852186665580312, outlierThresholdProb=0.03229809583062128, featureRetention=0.18594708334486176,
                     attitudeProcessVariance=6.025875320811407e-05, biasAccelProcessVariance=0.0, biasOmegaProcessVariance=0.0, cameraAttitudeProcessVariance=5.075382174045239e-06,
                     cameraPositionProcessVariance=1.2188313140115635e-05, pointProcessVariance=0.00029845436136043135, positionProcessVariance=9.981466095928483e-06,
                     velocityProcessVariance=0.025317333863551263, velAccNoise=0.012438843268295521, velAccBiasWalk=0.004462289865453429, velGyrNoise=0.000243153572917808,
                     velGyrBiasWalk=0.00013372703521098622, coordinateChoice=COORD_INVDEPTH, fastRiccati=1, useDiscreteInnovationLift=0, useDiscreteVelocityLift=1,
                     useEquivariantOutput=1, useFeaturePredictions=0, useMedianDepth=0).items():
        setattr(s, k, v)
    return s


def uzhfpv_settings():  # configs/EQVIO_config_UZHFPV.yaml:17-56
    s = Settings.defaults()
    for k, v in dict(initialSceneDepth=8.891397050194614, initialAttitudeVariance=0.10282752317467045, initialBiasAccelVariance=1.2232071190499316,
                     initialBiasOmegaVariance=1.1673134780260075, initialCameraAttitudeVariance=1.727825980507864e-07, initialCameraPositionVariance=3.349654391578276e-07,
                     initialPointVariance=100.0, initialPositionVariance=0.00011220184543019634, initialVelocityVariance=3.6517412725483775e-06,
                     measurementNoise=3.7583740428844425, outlierThresholdAbs=5.4509224619256385, outlierThresholdProb=0.23374912831534894, featureRetention=0.2,
                     attitudeProcessVariance=6.219421634147766e-08, biasAccelProcessVariance=0.0, biasOmegaProcessVariance=0.0, cameraAttitudeProcessVariance=2.2630153511576583e-06,
                     cameraPositionProcessVariance=6.853895838650084e-07, pointProcessVariance=0.000530103448340995, positionProcessVariance=1.2589961848499808e-05,
                     velocityProcessVariance=0.012232071190499315, velAccNoise=3.262345818455677e-05, velAccBiasWalk=0.0063404671195099425, velGyrNoise=0.0011913242870580211,
                     velGyrBiasWalk=0.00020008996495836354, coordinateChoice=COORD_INVDEPTH, fastRiccati=1, useDiscreteInnovationLift=0, useDiscreteVelocityLift=1,
                     useEquivariantOutput=1, useFeaturePredictions=0, useMedianDepth=0).items():
        setattr(s, k, v)
    return s


def sim_consistent(s, **kw):
    """The shipped dataset configs are tuned on real data (gyro-bias initial variance 9.7e4, outlier probability threshold
    0.03, ...): on the synthetic world they make the filter reject most features and diverge - in the oracle exactly as on
    the device - and cond(Sigma_0) = 1e12 puts the rounding floor at 1e-6. The stand-ins keep the dataset configs'
    structure (chart, lifts, fixed scene depth, process / velocity noise) and replace those values."""
    s.initialBiasOmegaVariance, s.initialBiasAccelVariance = 0.01, 0.01
    s.initialAttitudeVariance, s.initialPositionVariance, s.initialVelocityVariance = 1e-2, 1e-2, 1e-2
    s.initialCameraAttitudeVariance, s.initialCameraPositionVariance = 1e-4, 1e-4
    s.initialPointVariance = 4.0
    s.outlierThresholdAbs, s.outlierThresholdProb = 1e8, 1e8
    for k, v in kw.items():
        setattr(s, k, v)
    return s
