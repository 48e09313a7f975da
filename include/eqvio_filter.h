/* eqvio_filter.h — C-ABI over the host-side VIOFilter mirror (eqvio_amd/host/VIOFilter.hpp) for callers that are
 * not C++ (the Python tests and bench.py; a cgo/JNI/ctypes binding would use the same entry points).
 * Each function wraps exactly one public member of the reference's class VIOFilter
 * (include/eqvio/VIOFilter.h:86-192, src/VIOFilter.cpp). C++ callers use the class directly.
 *
 * Error convention: the reference has no return codes on this path (asserts, silent early returns). Here every
 * call returns 0 on success and -1 when the C++ layer threw (message via eqvio_filter_last_error); the silent
 * early returns of the reference (no IMU yet, stale stamp, empty measurement: VIOFilter.cpp:198-199, 223-224)
 * stay silent and return 0.
 */
#ifndef EQVIO_FILTER_H
#define EQVIO_FILTER_H
#include "eqf_hip.h"
#include "eqvio_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct eqvio_filter eqvio_filter;

/* VIOFilter(const Settings&) (src/VIOFilter.cpp:31-41) */
int eqvio_filter_create(eqvio_filter** out, const eqvio_settings* settings, int device, int max_landmarks);
/* VIOFilter(const VIOState&, const Settings&, const double& time) (src/VIOFilter.cpp:43-56) */
int eqvio_filter_create_from_state(eqvio_filter** out, const eqvio_settings* settings, int device, int max_landmarks, const double* sensor, const int* ids,
                                   const double* p, int N, double time);
void eqvio_filter_destroy(eqvio_filter* f);
const char* eqvio_filter_last_error(const eqvio_filter* f);

/* processIMUData (VIOFilter.cpp:58-63) / processVisionData (:194-241). ids ascending. */
int eqvio_filter_process_imu(eqvio_filter* f, const double* imu13);
int eqvio_filter_process_vision(eqvio_filter* f, double stamp, const eqvio_camera* cam, const int* ids, const double* y, int M);
/* stateEstimate (:243), getTime (:256), isInitialised (VIOFilter.h:167) */
int eqvio_filter_state_estimate(eqvio_filter* f, double* sensor, int* ids, double* p, int cap); /* returns N or -1 */
double eqvio_filter_get_time(const eqvio_filter* f);
int eqvio_filter_is_initialised(const eqvio_filter* f);
/* setState (:80-92), setLandmarks (:94-110), augmentLandmarkStates (:112-132) */
int eqvio_filter_set_state(eqvio_filter* f, const double* sensor, const int* ids, const double* p, int N);
int eqvio_filter_set_landmarks(eqvio_filter* f, const int* ids, const double* p, int N);
int eqvio_filter_augment_landmark_states(eqvio_filter* f, const int* new_ids, int n_new, const double* sensor, const int* ids, const double* p, int N);
/* viewEqFState() (:245): xi0, X, Sigma of the underlying VIO_eqf */
int eqvio_filter_get_eqf(eqvio_filter* f, double* xi0_sensor, double* X_sensor, int* ids, double* q0, double* Q, int cap); /* returns N or -1 */
int eqvio_filter_sigma_dim(const eqvio_filter* f);
int eqvio_filter_get_sigma(eqvio_filter* f, double* out_colmajor, int n);
/* viewEqFState().computeNEES(trueState) (src/main_sim.cpp:148, VIO_eqf.cpp:153-170) */
int eqvio_filter_compute_nees(eqvio_filter* f, const double* true_sensor, const int* true_ids, const double* true_p, int n_true, double* nees);
/* getFeaturePredictions(camPtr, stamp) (VIOFilter.cpp:247-252): ids ascending, pixel pairs; returns the count (0 unless
 * settings.useFeaturePredictions) or -1 */
int eqvio_filter_get_feature_predictions(eqvio_filter* f, const eqvio_camera* cam, double stamp, int* ids, double* y, int cap);
/* the device context behind viewEqFState(), for the eqf_* entry points */
eqf_ctx* eqvio_filter_core(eqvio_filter* f);
/* loopTimer sections of the last processVisionData (VIOFilter.cpp:196-236), seconds */
int eqvio_filter_last_timing(const eqvio_filter* f, double* propagation, double* preprocessing, double* correction);

/* Prepared replay. eqvio_frames_create builds, once, what the reference's tracker / data server hands to the filter: the IMU
 * samples and one VisionMeasurement (a std::map of pixel coordinates) per frame, from arrays concatenated over frames (frame j:
 * imu_counts[j] IMU samples of 13 doubles, then meas_counts[j] features at stamps[j]). eqvio_filter_run_prepared feeds frames
 * [first, first + count) to processIMUData / processVisionData and returns the number of frames processed or -1: it is what a
 * benchmark times, with the construction of the input containers outside the timed region. */
typedef struct eqvio_frames eqvio_frames;
eqvio_frames* eqvio_frames_create(const eqvio_camera* cam, int nframes, const int* imu_counts, const double* imu13_all, const double* stamps, const int* meas_counts,
                                  const int* ids_all, const double* y_all);
/* Overwrites the pixel of the k-th feature (ascending id order) of a prepared frame IN PLACE, through the measurement's public map and nothing else - what
 * a caller of the reference's VisionMeasurement type may do between building a measurement and handing it to the filter. The flat copies the host mirror
 * caches next to the map are validated against the map before every use that matters (tests/test_gpu_filter.py: an edited frame gives the same state as a
 * frame built with the new value). Returns 0, or -1 for a bad index. */
int eqvio_frames_edit_pixel(eqvio_frames* frames, int frame, int k, double u, double v);
int eqvio_frames_edit_id(eqvio_frames* frames, int frame, int k, int new_id); /* the k-th feature of the frame's std::map gets another id (erase + insert: same size) */
void eqvio_frames_destroy(eqvio_frames* frames);
int eqvio_frames_count(const eqvio_frames* frames);
int eqvio_filter_run_prepared(eqvio_filter* f, const eqvio_frames* frames, int first, int count);

/* Replay helper (create + run + destroy in one call): for each frame j, feed imu_counts[j] IMU samples (processIMUData) then one vision
 * measurement of meas_counts[j] features at stamps[j] (processVisionData). Arrays are concatenated over frames.
 * Returns the number of frames processed or -1. */
int eqvio_filter_run_frames(eqvio_filter* f, const eqvio_camera* cam, int nframes, const int* imu_counts, const double* imu13_all, const double* stamps,
                            const int* meas_counts, const int* ids_all, const double* y_all);

#ifdef __cplusplus
}
#endif
#endif
