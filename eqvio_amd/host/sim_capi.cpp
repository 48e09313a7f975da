// C-ABI wrapper of the synthetic-world data server (include/eqvio_sim.h).
#include "eqvio_sim.h"
#include "VIOSimulator.hpp"
#include <cstring>

using namespace eqvio_amd;

struct eqvio_sim {
    std::unique_ptr<SimulationDataServer> server;
};

extern "C" {

void eqvio_sim_default_settings(eqvio_sim_settings* s) {
    const SimSettings d;
    s->numPoints = d.numPoints;
    s->wallDistance = d.wallDistance;
    s->randomSeed = d.randomSeed;
    s->numWalls = d.numWalls;
    s->maxFeatures = (int)d.maxFeatures;
    s->initialNoise = s->inputNoise = s->outputNoise = 0;
    s->duration = d.duration;
    s->trajectory = EQVIO_TRAJ_WAVE;
    s->imuFreq = d.imuFreq;
    s->imageFreq = d.imageFreq;
}
eqvio_sim* eqvio_sim_create(const eqvio_sim_settings* s, const eqvio_settings* fs) {
    if (!s || !fs)
        return nullptr;
    try {
        SimSettings ss;
        ss.numPoints = s->numPoints;
        ss.wallDistance = s->wallDistance;
        ss.randomSeed = s->randomSeed;
        ss.numWalls = s->numWalls;
        ss.maxFeatures = (size_t)s->maxFeatures;
        ss.initialNoise = s->initialNoise != 0;
        ss.inputNoise = s->inputNoise != 0;
        ss.outputNoise = s->outputNoise != 0;
        ss.duration = s->duration;
        static const char* names[4] = {"wave", "square", "sine", "line"};
        ss.trajectory = names[(s->trajectory >= 0 && s->trajectory < 4) ? s->trajectory : 0];
        ss.imuFreq = s->imuFreq;
        ss.imageFreq = s->imageFreq;
        auto* h = new eqvio_sim();
        h->server = std::make_unique<SimulationDataServer>(ss, VIOFilter::Settings(*fs));
        return h;
    } catch (const std::exception&) {
        return nullptr;
    }
}
void eqvio_sim_destroy(eqvio_sim* s) { delete s; }
int eqvio_sim_next_measurement_type(const eqvio_sim* s) {
    switch (s->server->nextMeasurementType()) {
    case MeasurementType::Image: return EQVIO_MEAS_IMAGE;
    case MeasurementType::IMU: return EQVIO_MEAS_IMU;
    default: return EQVIO_MEAS_NONE;
    }
}
double eqvio_sim_next_time(const eqvio_sim* s) { return s->server->nextTime(); }
int eqvio_sim_get_imu(eqvio_sim* s, double* imu13) {
    s->server->getSimIMU().pack(imu13);
    return 0;
}
int eqvio_sim_get_vision(eqvio_sim* s, double* stamp, int* ids, double* y, int cap) {
    const VisionMeasurement m = s->server->getSimVision();
    *stamp = m.stamp;
    if ((int)m.camCoordinates.size() > cap)
        return -1;
    int k = 0;
    for (const auto& kv : m.camCoordinates) {
        ids[k] = kv.first;
        y[2 * k] = kv.second[0];
        y[2 * k + 1] = kv.second[1];
        ++k;
    }
    return k;
}
int eqvio_sim_true_state(const eqvio_sim* s, double stamp, int with_noise, double* sensor23, int* ids, double* p, int cap) {
    const VIOState xi = s->server->getTrueState(stamp, with_noise != 0);
    if ((int)xi.cameraLandmarks.size() > cap)
        return -1;
    std::memcpy(sensor23, xi.sensor.inputBias.data(), sizeof(double) * 6);
    const VIOSensorState& x = xi.sensor;
    const double v[17] = {x.pose.R.w, x.pose.R.x, x.pose.R.y, x.pose.R.z, x.pose.x.x, x.pose.x.y, x.pose.x.z, x.velocity.x, x.velocity.y, x.velocity.z,
                          x.cameraOffset.R.w, x.cameraOffset.R.x, x.cameraOffset.R.y, x.cameraOffset.R.z, x.cameraOffset.x.x, x.cameraOffset.x.y, x.cameraOffset.x.z};
    std::memcpy(sensor23 + 6, v, sizeof(v));
    for (size_t i = 0; i < xi.cameraLandmarks.size(); ++i) {
        ids[i] = xi.cameraLandmarks[i].id;
        p[3 * i] = xi.cameraLandmarks[i].p.x;
        p[3 * i + 1] = xi.cameraLandmarks[i].p.y;
        p[3 * i + 2] = xi.cameraLandmarks[i].p.z;
    }
    return (int)xi.cameraLandmarks.size();
}
int eqvio_sim_num_points(const eqvio_sim* s) { return (int)s->server->getTrueState(0.0, false).cameraLandmarks.size(); }
void eqvio_sim_camera(const eqvio_sim* s, eqvio_camera* cam) { *cam = s->server->viewSimulator().cameraPtr->c; }
void eqvio_sim_camera_offset(const eqvio_sim* s, double* q) {
    const Pose P = *s->server->cameraExtrinsics();
    const double v[7] = {P.R.w, P.R.x, P.R.y, P.R.z, P.x.x, P.x.y, P.x.z};
    std::memcpy(q, v, sizeof(v));
}
void eqvio_camera_project(const eqvio_camera* cam, const double* p3, double* y2) {
    Camera c;
    c.c = *cam;
    c.projectPoint(V3{p3[0], p3[1], p3[2]}, y2[0], y2[1]);
}
void eqvio_camera_undistort(const eqvio_camera* cam, const double* y2, double* b3) {
    Camera c;
    c.c = *cam;
    const V3 b = c.undistortPoint(y2[0], y2[1]);
    b3[0] = b.x;
    b3[1] = b.y;
    b3[2] = b.z;
}
void eqvio_camera_jacobian(const eqvio_camera* cam, const double* p3, double* J6) {
    Camera c;
    c.c = *cam;
    V3 j0, j1;
    eqf::cam_jac(c.model(), V3{p3[0], p3[1], p3[2]}, j0, j1);
    const double v[6] = {j0.x, j0.y, j0.z, j1.x, j1.y, j1.z};
    std::memcpy(J6, v, sizeof(v));
}
}
