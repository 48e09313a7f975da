// C-ABI wrapper of the host VIOFilter mirror (include/eqvio_filter.h).
#include "eqvio_filter.h"
#include "VIOFilter.hpp"
#include <cstring>

using namespace eqvio_amd;

struct eqvio_filter {
    std::unique_ptr<VIOFilter> filter;
    std::string err;
    double t_prop = 0, t_pre = 0, t_corr = 0;
};

namespace {
VIOState unpackState(const double* s, const int* ids, const double* p, int N) {
    VIOState xi;
    std::memcpy(xi.sensor.inputBias.data(), s, sizeof(double) * 6);
    xi.sensor.pose = Pose{Qt{s[6], s[7], s[8], s[9]}, V3{s[10], s[11], s[12]}};
    xi.sensor.velocity = V3{s[13], s[14], s[15]};
    xi.sensor.cameraOffset = Pose{Qt{s[16], s[17], s[18], s[19]}, V3{s[20], s[21], s[22]}};
    xi.cameraLandmarks.resize(N);
    for (int i = 0; i < N; ++i)
        xi.cameraLandmarks[i] = Landmark{V3{p[3 * i], p[3 * i + 1], p[3 * i + 2]}, ids[i]};
    return xi;
}
void packSensorOut(const VIOSensorState& x, double* s) {
    std::memcpy(s, x.inputBias.data(), sizeof(double) * 6);
    const double v[17] = {x.pose.R.w, x.pose.R.x, x.pose.R.y, x.pose.R.z, x.pose.x.x, x.pose.x.y, x.pose.x.z, x.velocity.x, x.velocity.y, x.velocity.z,
                          x.cameraOffset.R.w, x.cameraOffset.R.x, x.cameraOffset.R.y, x.cameraOffset.R.z, x.cameraOffset.x.x, x.cameraOffset.x.y, x.cameraOffset.x.z};
    std::memcpy(s + 6, v, sizeof(v));
}
IMUVelocity unpackIMU(const double* v) {
    IMUVelocity r;
    r.stamp = v[0];
    r.gyr = V3{v[1], v[2], v[3]};
    r.acc = V3{v[4], v[5], v[6]};
    r.gyrBiasVel = V3{v[7], v[8], v[9]};
    r.accBiasVel = V3{v[10], v[11], v[12]};
    return r;
}
VisionMeasurement makeMeasurement(double stamp, const GICameraPtr& cam, const int* ids, const double* y, int M) {
    VisionMeasurement m;
    m.stamp = stamp;
    m.cameraPtr = cam;
    for (int i = 0; i < M; ++i)
        m.camCoordinates[ids[i]] = {y[2 * i], y[2 * i + 1]};
    m.flatIds(); // flat form next to the map, built here with it
    return m;
}
GICameraPtr makeCamera(const eqvio_camera* c) {
    auto cam = std::make_shared<Camera>();
    cam->c = *c;
    return cam;
}
void initTimer() {
    // the labels the reference's mains initialise (src/main_opt.cpp:139-141, src/main_sim.cpp:81-82); loopTimer is
    // thread_local here, so every thread that drives a filter initialises its own once
    static thread_local bool done = false;
    if (done)
        return;
    done = true;
    loopTimer.initialise({"correction", "features", "preprocessing", "propagation", "total", "total vision update", "write output"});
}
template <typename F> int guarded(eqvio_filter* f, F&& fn) {
    try {
        fn();
        return 0;
    } catch (const std::exception& e) {
        if (f)
            f->err = e.what();
        return -1;
    }
}
} // namespace

extern "C" {

int eqvio_filter_create(eqvio_filter** out, const eqvio_settings* s, int device, int max_landmarks) {
    if (!out || !s)
        return -1;
    auto* f = new eqvio_filter();
    const int rc = guarded(f, [&] {
        VIOFilter::Settings st(*s);
        st.device = device;
        st.maxLandmarks = max_landmarks;
        initTimer();
        f->filter = std::make_unique<VIOFilter>(st);
    });
    *out = f;
    return rc;
}
int eqvio_filter_create_from_state(eqvio_filter** out, const eqvio_settings* s, int device, int max_landmarks, const double* sensor, const int* ids, const double* p,
                                   int N, double time) {
    if (!out || !s || !sensor)
        return -1;
    auto* f = new eqvio_filter();
    const int rc = guarded(f, [&] {
        VIOFilter::Settings st(*s);
        st.device = device;
        st.maxLandmarks = max_landmarks;
        initTimer();
        f->filter = std::make_unique<VIOFilter>(unpackState(sensor, ids, p, N), st, time);
    });
    *out = f;
    return rc;
}
void eqvio_filter_destroy(eqvio_filter* f) { delete f; }
const char* eqvio_filter_last_error(const eqvio_filter* f) { return f ? f->err.c_str() : "null filter"; }

int eqvio_filter_process_imu(eqvio_filter* f, const double* imu13) {
    return guarded(f, [&] { f->filter->processIMUData(unpackIMU(imu13)); });
}
static void grabTiming(eqvio_filter* f) {
    const auto& t = loopTimer.getLoopTimingData().timings;
    auto get = [&t](const char* k) {
        const auto it = t.find(k);
        return it == t.end() ? 0.0 : it->second.count();
    };
    f->t_prop = get("propagation");
    f->t_pre = get("preprocessing");
    f->t_corr = get("correction");
}
int eqvio_filter_process_vision(eqvio_filter* f, double stamp, const eqvio_camera* cam, const int* ids, const double* y, int M) {
    return guarded(f, [&] {
        initTimer();
        f->filter->processVisionData(makeMeasurement(stamp, makeCamera(cam), ids, y, M));
        grabTiming(f);
    });
}
int eqvio_filter_state_estimate(eqvio_filter* f, double* sensor, int* ids, double* p, int cap) {
    int N = -1;
    const int rc = guarded(f, [&] {
        const VIOState xi = f->filter->stateEstimate();
        if ((int)xi.cameraLandmarks.size() > cap)
            throw std::length_error("state_estimate: capacity");
        if (sensor)
            packSensorOut(xi.sensor, sensor);
        N = (int)xi.cameraLandmarks.size();
        for (int i = 0; i < N; ++i) {
            if (ids)
                ids[i] = xi.cameraLandmarks[i].id;
            if (p) {
                p[3 * i] = xi.cameraLandmarks[i].p.x;
                p[3 * i + 1] = xi.cameraLandmarks[i].p.y;
                p[3 * i + 2] = xi.cameraLandmarks[i].p.z;
            }
        }
    });
    return rc ? -1 : N;
}
double eqvio_filter_get_time(const eqvio_filter* f) { return f->filter->getTime(); }
int eqvio_filter_is_initialised(const eqvio_filter* f) { return f->filter->isInitialised() ? 1 : 0; }
int eqvio_filter_set_state(eqvio_filter* f, const double* sensor, const int* ids, const double* p, int N) {
    return guarded(f, [&] { f->filter->setState(unpackState(sensor, ids, p, N)); });
}
int eqvio_filter_set_landmarks(eqvio_filter* f, const int* ids, const double* p, int N) {
    const double s0[23] = {0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0};
    return guarded(f, [&] { f->filter->setLandmarks(unpackState(s0, ids, p, N).cameraLandmarks); });
}
int eqvio_filter_augment_landmark_states(eqvio_filter* f, const int* new_ids, int n_new, const double* sensor, const int* ids, const double* p, int N) {
    return guarded(f, [&] { f->filter->augmentLandmarkStates(std::vector<int>(new_ids, new_ids + n_new), unpackState(sensor, ids, p, N)); });
}
int eqvio_filter_get_eqf(eqvio_filter* f, double* xi0_sensor, double* X_sensor, int* ids, double* q0, double* Q, int cap) {
    const int rc = eqf_get_state(f->filter->eqfState().ctx, xi0_sensor, X_sensor, ids, q0, Q, cap);
    return rc < 0 ? -1 : rc;
}
int eqvio_filter_sigma_dim(const eqvio_filter* f) { return 21 + 3 * f->filter->viewEqFState().numLandmarks(); }
int eqvio_filter_get_sigma(eqvio_filter* f, double* out, int n) { return eqf_get_sigma(f->filter->eqfState().ctx, out, n) == 0 ? 0 : -1; }
int eqvio_filter_compute_nees(eqvio_filter* f, const double* ts, const int* tids, const double* tp, int nt, double* nees) {
    return guarded(f, [&] { *nees = f->filter->viewEqFState().computeNEES(unpackState(ts, tids, tp, nt)); });
}
int eqvio_filter_get_feature_predictions(eqvio_filter* f, const eqvio_camera* cam, double stamp, int* ids, double* y, int cap) {
    int count = -1;
    const int rc = guarded(f, [&] {
        const VisionMeasurement m = f->filter->getFeaturePredictions(makeCamera(cam), stamp);
        if ((int)m.camCoordinates.size() > cap)
            throw std::runtime_error("eqvio_filter_get_feature_predictions: capacity");
        int k = 0;
        for (const auto& kv : m.camCoordinates) {
            ids[k] = kv.first;
            y[2 * k] = kv.second[0];
            y[2 * k + 1] = kv.second[1];
            ++k;
        }
        count = k;
    });
    return rc ? -1 : count;
}
eqf_ctx* eqvio_filter_core(eqvio_filter* f) { return f->filter->eqfState().ctx; }
int eqvio_filter_last_timing(const eqvio_filter* f, double* a, double* b, double* c) {
    if (a)
        *a = f->t_prop;
    if (b)
        *b = f->t_pre;
    if (c)
        *c = f->t_corr;
    return 0;
}
// Prepared replay: the IMU samples and the VisionMeasurement objects (a std::map per frame, as the reference's tracker / data
// server hands them to the filter, main_opt.cpp:196-214) are built once, outside any timed region.
struct eqvio_frames {
    GICameraPtr camPtr;
    std::vector<VisionMeasurement> meas;
    std::vector<IMUVelocity> imus;
    std::vector<size_t> imuBegin; // nframes + 1 offsets into imus
};
eqvio_frames* eqvio_frames_create(const eqvio_camera* cam, int nframes, const int* imu_counts, const double* imu13_all, const double* stamps, const int* meas_counts,
                                  const int* ids_all, const double* y_all) {
    if (!cam || nframes < 0 || (nframes > 0 && (!imu_counts || !stamps || !meas_counts)))
        return nullptr;
    try {
        auto* fr = new eqvio_frames;
        fr->camPtr = makeCamera(cam);
        fr->meas.resize((size_t)nframes);
        fr->imuBegin.assign(1, 0);
        size_t io = 0, mo = 0;
        for (int j = 0; j < nframes; ++j) {
            for (int s = 0; s < imu_counts[j]; ++s)
                fr->imus.push_back(unpackIMU(imu13_all + 13 * (io + s)));
            io += imu_counts[j];
            fr->imuBegin.push_back(io);
            fr->meas[j] = makeMeasurement(stamps[j], fr->camPtr, ids_all + mo, y_all + 2 * mo, meas_counts[j]);
            mo += meas_counts[j];
        }
        return fr;
    } catch (...) {
        return nullptr;
    }
}
int eqvio_frames_edit_pixel(eqvio_frames* fr, int frame, int k, double u, double v) {
    // what a caller holding the reference's VisionMeasurement may do: write into the public std::map, nothing else
    if (!fr || frame < 0 || (size_t)frame >= fr->meas.size() || k < 0 || (size_t)k >= fr->meas[frame].camCoordinates.size())
        return -1;
    auto it = fr->meas[frame].camCoordinates.begin();
    std::advance(it, k);
    it->second = {u, v};
    return 0;
}
int eqvio_frames_edit_id(eqvio_frames* fr, int frame, int k, int new_id) {
    // ... or replace a feature: erase the k-th entry of the public std::map and insert its pixel under another id (the size stays: a cached copy cannot tell by it)
    if (!fr || frame < 0 || (size_t)frame >= fr->meas.size() || k < 0 || (size_t)k >= fr->meas[frame].camCoordinates.size())
        return -1;
    auto& m = fr->meas[frame].camCoordinates;
    if (m.count(new_id))
        return -1;
    auto it = m.begin();
    std::advance(it, k);
    const auto px = it->second;
    m.erase(it);
    m[new_id] = px;
    return 0;
}
void eqvio_frames_destroy(eqvio_frames* fr) { delete fr; }
int eqvio_frames_count(const eqvio_frames* fr) { return fr ? (int)fr->meas.size() : -1; }
int eqvio_filter_run_prepared(eqvio_filter* f, const eqvio_frames* fr, int first, int count) {
    if (!fr || first < 0 || count < 0 || (size_t)first + (size_t)count > fr->meas.size())
        return -1;
    int done = 0;
    const int rc = guarded(f, [&] {
        initTimer();
        for (int j = first; j < first + count; ++j) {
            for (size_t s = fr->imuBegin[j]; s < fr->imuBegin[j + 1]; ++s)
                f->filter->processIMUData(fr->imus[s]);
            f->filter->processVisionData(fr->meas[j]);
            ++done;
        }
        grabTiming(f);
    });
    return rc ? -1 : done;
}
int eqvio_filter_run_frames(eqvio_filter* f, const eqvio_camera* cam, int nframes, const int* imu_counts, const double* imu13_all, const double* stamps,
                            const int* meas_counts, const int* ids_all, const double* y_all) {
    eqvio_frames* fr = eqvio_frames_create(cam, nframes, imu_counts, imu13_all, stamps, meas_counts, ids_all, y_all);
    if (!fr)
        return -1;
    const int done = eqvio_filter_run_prepared(f, fr, 0, nframes);
    eqvio_frames_destroy(fr);
    return done;
}

} // extern "C"
