/* Plain-C value types shared by the C-ABI entry points of the MI355X EqF path.
 *
 * They flatten the reference's C++ value types (which depend on Eigen / LiePP / GIFT) into plain
 * doubles and ints so that the boundary carries no library types:
 *
 *   VIOSensorState (include/eqvio/mathematical/VIOState.h:57-71)    -> double[23]  EQVIO_SENSOR_DIM
 *       [0:6)  inputBias (gyr, acc)
 *       [6:10) pose.R   quaternion (w,x,y,z)    [10:13) pose.x
 *       [13:16) velocity (body frame)
 *       [16:20) cameraOffset.R quaternion (w,x,y,z)   [20:23) cameraOffset.x
 *   VIOGroup sensor part (include/eqvio/mathematical/VIOGroup.h:32-38)  -> double[23]
 *       [0:6) beta   [6:10) A.R (w,x,y,z)  [10:13) A.x   [13:16) w   [16:20) B.R (w,x,y,z)  [20:23) B.x
 *   Landmark (VIOState.h:41-46)                -> int id ; double p[3]
 *   liepp::SOT3d Q_i (VIOGroup.h:37)           -> double[5] = (qw,qx,qy,qz,a)
 *   IMUVelocity (IMUVelocity.h:33-84)          -> double[13] = stamp, gyr[3], acc[3], gyrBiasVel[3], accBiasVel[3]
 *   VisionMeasurement (VisionMeasurement.h:35-62) -> stamp, ids[M] ASCENDING, y[2M] pixel coordinates, camera
 *   GIFT::GICameraPtr                           -> eqvio_camera
 *   VIOFilter::Settings (include/eqvio/VIOFilterSettings.h:58-124) -> eqvio_settings (same field names)
 *
 * Quaternion order (w,x,y,z) follows the reference's CSV/YAML convention
 * (include/eqvio/csv/CSVLine.h:204-248, include/eqvio/common/LieYaml.h:42-57).
 */
#ifndef EQVIO_TYPES_H
#define EQVIO_TYPES_H

#ifdef __cplusplus
extern "C" {
#endif

#define EQVIO_SENSOR_DIM 23
#define EQVIO_IMU_DIM 13

enum { EQVIO_COORD_EUCLIDEAN = 0, EQVIO_COORD_INVDEPTH = 1, EQVIO_COORD_NORMAL = 2 };
/* Camera models the reference's dataset readers construct (the classes live in the GIFT submodule, external/GIFT,
 * which is not vendored in /root/reference: the models are restated from their published definitions):
 *   PINHOLE      GIFT::PinholeCamera      (SimulationDataServer.cpp:162-176)
 *   RADTAN       GIFT::StandardCamera     pinhole + radial-tangential, dist = (k1, k2, p1, p2, k3), the OpenCV order of
 *                                         intrinsics.yaml:8 / sensor.yaml distortion_coefficients (ASLDatasetReader.cpp:90-94)
 *   EQUIDISTANT  GIFT::EquidistantCamera  Kannala-Brandt, dist = (k1, k2, k3, k4)  (UZHFPVDatasetReader.cpp:99-102) */
enum { EQVIO_CAMERA_PINHOLE = 0, EQVIO_CAMERA_RADTAN = 1, EQVIO_CAMERA_EQUIDISTANT = 2 };

typedef struct eqvio_camera {
    int model; /* EQVIO_CAMERA_* */
    int width, height;
    double fx, fy, cx, cy;
    double dist[5];
} eqvio_camera;

/* Field names and defaults follow VIOFilter::Settings (VIOFilterSettings.h:59-99). */
typedef struct eqvio_settings {
    double biasOmegaProcessVariance, biasAccelProcessVariance, attitudeProcessVariance, positionProcessVariance,
        velocityProcessVariance, cameraAttitudeProcessVariance, cameraPositionProcessVariance, pointProcessVariance;
    double velGyrNoise, velAccNoise, velGyrBiasWalk, velAccBiasWalk;
    double measurementNoise, outlierThresholdAbs, outlierThresholdProb, featureRetention;
    double initialAttitudeVariance, initialPositionVariance, initialVelocityVariance, initialCameraAttitudeVariance,
        initialCameraPositionVariance, initialPointVariance, initialPointDepthVariance, initialBiasOmegaVariance,
        initialBiasAccelVariance, initialSceneDepth;
    int useDiscreteInnovationLift, useDiscreteVelocityLift, useDiscreteStateMatrix, fastRiccati, useMedianDepth,
        useFeaturePredictions, useEquivariantOutput, removeLostLandmarks;
    int coordinateChoice;   /* EQVIO_COORD_* */
    double cameraOffset[7]; /* (qw,qx,qy,qz, x,y,z) */
} eqvio_settings;

#ifdef __cplusplus
}
#endif
#endif
