/* C view of the synthetic-world data server for the MI355X EqF path (eqvio_amd/host/VIOSimulator.hpp), the caller
 * side of the hot path: SURVEY.md §8 row f-1.
 * Replaces, for non-C++ callers, the reference's
 *   class SimulationDataServer   include/eqvio/dataserver/SimulationDataServer.h:25-70
 *   class VIOSimulator           include/eqvio/VIOSimulator.h:29-106
 * Host-only (no device work). Exported by eqvio_amd/lib/libeqvio_filter.so. */
#ifndef EQVIO_SIM_H
#define EQVIO_SIM_H
#include "eqvio_types.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct eqvio_sim eqvio_sim;

enum { EQVIO_TRAJ_WAVE = 0, EQVIO_TRAJ_SQUARE = 1, EQVIO_TRAJ_SINE = 2, EQVIO_TRAJ_LINE = 3 };
enum { EQVIO_MEAS_IMAGE = 0, EQVIO_MEAS_IMU = 1, EQVIO_MEAS_NONE = 2 }; /* MeasurementType, DataServerBase.h */

/* the "sim:" configuration block (VIOSimulator.cpp:47-62, SimulationDataServer.cpp:222-231) */
typedef struct {
    int numPoints;       /* 1000 */
    double wallDistance; /* 2.0 */
    unsigned randomSeed;
    int numWalls;    /* 1 */
    int maxFeatures; /* 30 */
    int initialNoise, inputNoise, outputNoise;
    double duration; /* 100 s */
    int trajectory;  /* EQVIO_TRAJ_* */
    double imuFreq, imageFreq; /* 200, 20 Hz */
} eqvio_sim_settings;

void eqvio_sim_default_settings(eqvio_sim_settings* s);
/* SimulationDataServer(simSettings, filterSettings) */
eqvio_sim* eqvio_sim_create(const eqvio_sim_settings* sim, const eqvio_settings* filter_settings);
void eqvio_sim_destroy(eqvio_sim* s);
int eqvio_sim_next_measurement_type(const eqvio_sim* s); /* nextMeasurementType */
double eqvio_sim_next_time(const eqvio_sim* s);          /* nextTime (NaN at the end) */
int eqvio_sim_get_imu(eqvio_sim* s, double* imu13);      /* getSimIMU */
/* getSimVision: writes the stamp, ascending ids and pixel pairs; returns the feature count (or -1 if cap is too small) */
int eqvio_sim_get_vision(eqvio_sim* s, double* stamp, int* ids, double* y, int cap);
/* getTrueState(stamp, withNoise): sensor[23], landmark ids and camera-frame positions of ALL world points in world
 * order; returns the landmark count (or -1 if cap is too small) */
int eqvio_sim_true_state(const eqvio_sim* s, double stamp, int with_noise, double* sensor23, int* ids, double* p, int cap);
int eqvio_sim_num_points(const eqvio_sim* s);
void eqvio_sim_camera(const eqvio_sim* s, eqvio_camera* cam);    /* generatePinholeCameraSquare */
void eqvio_sim_camera_offset(const eqvio_sim* s, double* pose7); /* cameraExtrinsics, (qw,qx,qy,qz,x,y,z) */


/* The camera-model functions of the product (eqvio_amd/csrc/eqf_math.hpp: the same code the kernels run), callable on the
 * host: GIFT::GICamera::projectPoint / undistortPoint (unit bearing) / projectionJacobian (2 x 3, row major). */
void eqvio_camera_project(const eqvio_camera* cam, const double* p3, double* y2);
void eqvio_camera_undistort(const eqvio_camera* cam, const double* y2, double* bearing3);
void eqvio_camera_jacobian(const eqvio_camera* cam, const double* p3, double* J6);

#ifdef __cplusplus
}
#endif
#endif
