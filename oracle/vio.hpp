// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build/link/call this.
//
// CPU restatement (plain C++, fp64, no Eigen/LiePP/GIFT) of the EqVIO EqF hot path:
// geometry + symmetry group + coordinate charts + EqF matrices + the EqF itself + the VIOFilter wrapper.
// Every function cites the reference file:line it follows (paths relative to /root/reference).
//
// PARITY PINNING STATUS: the reference holds NO golden vectors / known-answer fixtures (all of its
// tests are randomised property tests) and the reference itself cannot be built in this
// image (Eigen, LiePP, GIFT, OpenCV, yaml-cpp, gtest absent; external/ submodules empty; no network), so
// outputs of the reference are not available: in the strict sense of "pinned by the reference's own
// vectors or outputs" this oracle is PARITY UNPINNED, and DESIGN.md §5 says so. What pins it instead:
//  (1) the reference's property tests restated against it (oracle/prop_tests.cpp, run by
//      tests/test_oracle_properties.py): group axioms, action compatibility, output equivariance,
//      discrete-lift exactness (1e-12), A/B/C vs numerical differentials, invdepth = M*euclid*M^-1, chart
//      round trips, innovation-lift identities and the NEES statistics test;
//  (2) a SECOND restatement written independently from the reference's sources (oracle/indep/eqvio_ref.py:
//      numpy, rotation matrices instead of quaternions, its own exp / log / Jacobian formulas): every golden
//      output is regenerated from the golden inputs with it and the two fixture families must agree
//      (tests/test_golden.py, 1e-12 on the analytic paths);
//  (3) the same second restatement evaluated with 50 digits (mpmath) as the truth for the dense
//      Sigma / K / Gamma arithmetic of performVisionUpdate (tests/test_truth_mp.py).
#pragma once
#include "lie.hpp"
#include <functional>
#include <map>
#include <memory>
#include <numeric>
#include <set>
#include <string>

namespace orc {

constexpr double GRAVITY_CONSTANT = 9.80665; // include/eqvio/mathematical/IMUVelocity.h:26

// ---------------------------------------------------------------- camera (GIFT camera models; source absent)
// The classes live in the GIFT submodule (external/GIFT, github.com/pvangoor/GIFT, commit not recorded in this snapshot),
// which is NOT in /root/reference: restated from the call-site contract and the published model
// definitions. project(p) = K distort(x/z, y/z) + c, undistortPoint(y) = unit bearing, projectionJacobian(p) = d project/d p.
//   model 0  pinhole                (GIFT::PinholeCamera,     SimulationDataServer.cpp:162-176)
//   model 1  radial-tangential      (GIFT::StandardCamera,    ASLDatasetReader.cpp:90-94; OpenCV order k1 k2 p1 p2 k3)
//   model 2  equidistant            (GIFT::EquidistantCamera, UZHFPVDatasetReader.cpp:99-102; Kannala-Brandt k1..k4)
// PARITY: the inverse (undistortPoint) is a function, not an algorithm: any fully converged iteration agrees to rounding.
// Here: the classic fixed-point iteration followed by Newton polishing (the product uses Newton from the start).
struct Camera {
    int model = 0;
    double fx = 1, fy = 1, cx = 0, cy = 0;
    int width = 0, height = 0;
    double dist[5] = {0, 0, 0, 0, 0};
    // radial factor and tangential offset of the radtan model at normalised (x, y)
    void radtanTerms(double x, double y, double& radial, double& tx, double& ty) const {
        const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
        radial = 1.0 + dist[0] * r2 + dist[1] * r4 + dist[4] * r6;
        tx = 2.0 * dist[2] * x * y + dist[3] * (r2 + 2.0 * x * x);
        ty = dist[2] * (r2 + 2.0 * y * y) + 2.0 * dist[3] * x * y;
    }
    double kbTheta(double th) const { // theta_d(theta)
        const double t2 = th * th, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
        return th * (1.0 + dist[0] * t2 + dist[1] * t4 + dist[2] * t6 + dist[3] * t8);
    }
    double kbThetaDiff(double th) const {
        const double t2 = th * th, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
        return 1.0 + 3.0 * dist[0] * t2 + 5.0 * dist[1] * t4 + 7.0 * dist[2] * t6 + 9.0 * dist[3] * t8;
    }
    Vec2 distort(const Vec2& xy) const {
        if (model == 1) {
            double radial, tx, ty;
            radtanTerms(xy(0), xy(1), radial, tx, ty);
            return vec2(xy(0) * radial + tx, xy(1) * radial + ty);
        }
        if (model == 2) {
            const double r = std::sqrt(xy(0) * xy(0) + xy(1) * xy(1));
            if (r < 1e-8)
                return xy;
            return xy * (kbTheta(std::atan(r)) / r);
        }
        return xy;
    }
    M<2, 2> distortJacobian(const Vec2& xy) const {
        M<2, 2> J = M<2, 2>::Identity();
        const double x = xy(0), y = xy(1);
        if (model == 1) {
            const double k1 = dist[0], k2 = dist[1], p1 = dist[2], p2 = dist[3], k3 = dist[4];
            const double r2 = x * x + y * y;
            const double radial = 1.0 + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2;
            const double dRadial = k1 + 2.0 * k2 * r2 + 3.0 * k3 * r2 * r2; // d radial / d r2
            J(0, 0) = radial + x * dRadial * 2.0 * x + 2.0 * p1 * y + p2 * (2.0 * x + 4.0 * x);
            J(0, 1) = x * dRadial * 2.0 * y + 2.0 * p1 * x + p2 * 2.0 * y;
            J(1, 0) = y * dRadial * 2.0 * x + p1 * 2.0 * x + 2.0 * p2 * y;
            J(1, 1) = radial + y * dRadial * 2.0 * y + p1 * (2.0 * y + 4.0 * y) + 2.0 * p2 * x;
        } else if (model == 2) {
            const double r = std::sqrt(x * x + y * y);
            if (r >= 1e-8) {
                const double th = std::atan(r);
                const double s = kbTheta(th) / r;
                const double ds = (kbThetaDiff(th) / (1.0 + r * r) * r - kbTheta(th)) / (r * r); // d s / d r
                J(0, 0) = s + ds * x * x / r;
                J(0, 1) = ds * x * y / r;
                J(1, 0) = J(0, 1);
                J(1, 1) = s + ds * y * y / r;
            }
        }
        return J;
    }
    Vec2 projectPoint(const Vec3& p) const {
        const Vec2 d = distort(vec2(p(0) / p(2), p(1) / p(2)));
        return vec2(fx * d(0) + cx, fy * d(1) + cy);
    }
    Vec3 undistortPoint(const Vec2& y) const {
        const Vec2 d = vec2((y(0) - cx) / fx, (y(1) - cy) / fy);
        if (model == 0)
            return vec3(d(0), d(1), 1.0).normalized();
        if (model == 2) {
            const double thd = std::sqrt(d(0) * d(0) + d(1) * d(1));
            if (thd < 1e-8)
                return vec3(d(0), d(1), 1.0).normalized();
            double th = thd;
            for (int it = 0; it < 30; ++it) // bisection-safe Newton: theta_d is monotone on the image
                th -= (kbTheta(th) - thd) / kbThetaDiff(th);
            const double r = std::tan(th);
            return vec3(d(0) * r / thd, d(1) * r / thd, 1.0).normalized();
        }
        Vec2 xy = d;
        for (int it = 0; it < 20; ++it) { // x <- (x_d - tangential(x)) / radial(x)
            double radial, tx, ty;
            radtanTerms(xy(0), xy(1), radial, tx, ty);
            xy = vec2((d(0) - tx) / radial, (d(1) - ty) / radial);
        }
        for (int it = 0; it < 4; ++it) { // Newton polish to rounding
            const Vec2 e = distort(xy) - d;
            const M<2, 2> J = distortJacobian(xy);
            const double det = J(0, 0) * J(1, 1) - J(0, 1) * J(1, 0);
            xy = vec2(xy(0) - (J(1, 1) * e(0) - J(0, 1) * e(1)) / det, xy(1) - (J(0, 0) * e(1) - J(1, 0) * e(0)) / det);
        }
        return vec3(xy(0), xy(1), 1.0).normalized();
    }
    M<2, 3> projectionJacobian(const Vec3& p) const {
        const double iz = 1.0 / p(2);
        M<2, 3> Jn = M<2, 3>::Zero(); // d (x/z, y/z) / d p
        Jn(0, 0) = iz;
        Jn(0, 2) = -p(0) * iz * iz;
        Jn(1, 1) = iz;
        Jn(1, 2) = -p(1) * iz * iz;
        M<2, 2> Kf = M<2, 2>::Zero();
        Kf(0, 0) = fx;
        Kf(1, 1) = fy;
        return Kf * distortJacobian(vec2(p(0) * iz, p(1) * iz)) * Jn;
    }
    bool isInDomain(const Vec3& p) const {
        if (p(2) <= 0)
            return false;
        const Vec2 y = projectPoint(p);
        return y(0) >= 0 && y(0) < width && y(1) >= 0 && y(1) < height;
    }
};
using CameraPtr = std::shared_ptr<const Camera>;

// ---------------------------------------------------------------- IMU velocity (src/mathematical/IMUVelocity.cpp)
struct IMUVelocity {
    double stamp = 0;
    Vec3 gyr = Vec3::Zero(), acc = Vec3::Zero(), gyrBiasVel = Vec3::Zero(), accBiasVel = Vec3::Zero();
    static IMUVelocity Zero() { return IMUVelocity(); }                 // IMUVelocity.cpp:19-25
    IMUVelocity operator+(const IMUVelocity& o) const {                 // :42-50
        IMUVelocity r;
        r.stamp = (stamp > 0) ? stamp : o.stamp;
        r.gyr = gyr + o.gyr;
        r.acc = acc + o.acc;
        r.gyrBiasVel = gyrBiasVel + o.gyrBiasVel;
        r.accBiasVel = accBiasVel + o.accBiasVel;
        return r;
    }
    IMUVelocity minusBias(const Vec6& b) const {                        // :52-58 (bias velocities come out ZERO)
        IMUVelocity r;
        r.stamp = stamp;
        r.gyr = gyr - b.block<3, 1>(0, 0);
        r.acc = acc - b.block<3, 1>(3, 0);
        return r;
    }
    IMUVelocity operator*(double c) const {                             // :69-77
        IMUVelocity r;
        r.stamp = stamp;
        r.gyr = gyr * c;
        r.acc = acc * c;
        r.gyrBiasVel = gyrBiasVel * c;
        r.accBiasVel = accBiasVel * c;
        return r;
    }
    static IMUVelocity fromVec12(const M<12, 1>& v) {                   // :33-40
        IMUVelocity r;
        r.gyr = v.block<3, 1>(0, 0);
        r.acc = v.block<3, 1>(3, 0);
        r.gyrBiasVel = v.block<3, 1>(6, 0);
        r.accBiasVel = v.block<3, 1>(9, 0);
        return r;
    }
};

// ---------------------------------------------------------------- vision measurement (src/mathematical/VisionMeasurement.cpp)
struct VisionMeasurement {
    double stamp = 0;
    std::map<int, Vec2> camCoordinates;
    CameraPtr cameraPtr;
    std::vector<int> getIds() const {
        std::vector<int> ids;
        for (const auto& kv : camCoordinates)
            ids.push_back(kv.first);
        return ids;
    }
    DVec asVector() const { // operator VectorXd, :72-79 (ascending id order)
        DVec v;
        for (const auto& kv : camCoordinates) {
            v.push_back(kv.second(0));
            v.push_back(kv.second(1));
        }
        return v;
    }
};
inline VisionMeasurement operator-(const VisionMeasurement& y1, const VisionMeasurement& y2) { // :60-71
    VisionMeasurement d;
    for (const auto& kv : y1.camCoordinates) {
        const auto it = y2.camCoordinates.find(kv.first);
        if (it != y2.camCoordinates.end())
            d.camCoordinates[kv.first] = kv.second - it->second;
    }
    d.cameraPtr = y1.cameraPtr;
    return d;
}
inline VisionMeasurement operator+(const VisionMeasurement& y, const DVec& eta) { // :81-89
    VisionMeasurement r = y;
    size_t i = 0;
    for (auto& kv : r.camCoordinates) {
        kv.second = kv.second + vec2(eta[2 * i], eta[2 * i + 1]);
        ++i;
    }
    return r;
}

// ---------------------------------------------------------------- state (include/eqvio/mathematical/VIOState.h:41-90)
struct Landmark {
    Vec3 p = Vec3::Zero();
    int id = -1;
};
struct VIOSensorState {
    Vec6 inputBias = Vec6::Zero();
    SE3 pose;
    Vec3 velocity = Vec3::Zero();
    SE3 cameraOffset;
    Vec3 gravityDir() const { return pose.R.inverse() * vec3(0, 0, 1); } // VIOState.cpp:94
    static constexpr int CompDim = 21;
};
struct VIOState {
    VIOSensorState sensor;
    std::vector<Landmark> cameraLandmarks;
    std::vector<int> getIds() const {
        std::vector<int> ids;
        for (const auto& lm : cameraLandmarks)
            ids.push_back(lm.id);
        return ids;
    }
    int Dim() const { return VIOSensorState::CompDim + 3 * (int)cameraLandmarks.size(); }
};

// VIOState.cpp:28-68
inline VIOState integrateSystemFunction(const VIOState& state, const IMUVelocity& velocity, double dt) {
    VIOState ns;
    const IMUVelocity v_est = velocity.minusBias(state.sensor.inputBias);
    ns.sensor.inputBias.setBlock<3, 1>(0, 0, state.sensor.inputBias.block<3, 1>(0, 0) + dt * velocity.gyrBiasVel);
    ns.sensor.inputBias.setBlock<3, 1>(3, 0, state.sensor.inputBias.block<3, 1>(3, 0) + dt * velocity.accBiasVel);
    const VIOSensorState& s = state.sensor;
    SE3 poseChange;
    poseChange.R = SO3::exp(dt * v_est.gyr);
    poseChange.x = dt * (s.pose.R * s.velocity) + 0.5 * dt * dt * (s.pose.R * v_est.acc + vec3(0, 0, -GRAVITY_CONSTANT));
    poseChange.x = s.pose.R.inverse() * poseChange.x;
    ns.sensor.pose = s.pose * poseChange;
    const Vec3 inertialVelocityDiff = s.pose.R.asMatrix() * v_est.acc + vec3(0, 0, -GRAVITY_CONSTANT);
    ns.sensor.velocity = ns.sensor.pose.R.inverse() * (s.pose.R * s.velocity + dt * inertialVelocityDiff);
    const SE3 cameraPoseChangeInv = s.cameraOffset.inverse() * poseChange.inverse() * s.cameraOffset;
    ns.cameraLandmarks.resize(state.cameraLandmarks.size());
    for (size_t i = 0; i < state.cameraLandmarks.size(); ++i) {
        ns.cameraLandmarks[i].p = cameraPoseChangeInv * state.cameraLandmarks[i].p;
        ns.cameraLandmarks[i].id = state.cameraLandmarks[i].id;
    }
    ns.sensor.cameraOffset = s.cameraOffset;
    return ns;
}
// VIOState.cpp:70-78
inline VisionMeasurement measureSystemState(const VIOState& state, const CameraPtr& cam) {
    VisionMeasurement r;
    for (const auto& lm : state.cameraLandmarks)
        r.camCoordinates.insert({lm.id, cam->projectPoint(lm.p)});
    r.cameraPtr = cam;
    return r;
}

// ---------------------------------------------------------------- symmetry group (include/eqvio/mathematical/VIOGroup.h:32-119)
struct VIOGroup {
    Vec6 beta = Vec6::Zero();
    SE3 A;
    Vec3 w = Vec3::Zero();
    SE3 B;
    std::vector<SOT3> Q;
    std::vector<int> id;
    static VIOGroup Identity(const std::vector<int>& ids = {}) { // VIOGroup.cpp:94-106
        VIOGroup X;
        X.id = ids;
        X.Q.assign(ids.size(), SOT3::Identity());
        return X;
    }
    VIOGroup operator*(const VIOGroup& o) const { // :71-92
        VIOGroup r;
        r.beta = beta + o.beta;
        r.A = A * o.A;
        r.B = B * o.B;
        r.w = w + A.R * o.w;
        assert(Q.size() == o.Q.size());
        r.Q.resize(Q.size());
        for (size_t i = 0; i < Q.size(); ++i)
            r.Q[i] = Q[i] * o.Q[i];
        r.id = id;
        return r;
    }
    VIOGroup inverse() const { // :108-120
        VIOGroup r;
        r.beta = -beta;
        r.A = A.inverse();
        r.B = B.inverse();
        r.w = -(A.R.inverse() * w);
        r.Q = Q;
        for (auto& q : r.Q)
            q = q.inverse();
        r.id = id;
        return r;
    }
    bool hasNaN() const {
        bool f = beta.hasNaN() || A.hasNaN() || B.hasNaN() || w.hasNaN();
        for (const auto& q : Q)
            f = f || q.hasNaN();
        return f;
    }
};
struct VIOAlgebra {
    Vec6 u_beta = Vec6::Zero();
    Vec6 U_A = Vec6::Zero();
    Vec6 U_B = Vec6::Zero();
    Vec3 u_w = Vec3::Zero();
    std::vector<Vec4> W;
    std::vector<int> id;
    VIOAlgebra operator*(double c) const { // VIOGroup.cpp:142-153
        VIOAlgebra r = *this;
        r.u_beta = u_beta * c;
        r.U_A = U_A * c;
        r.U_B = U_B * c;
        r.u_w = u_w * c;
        for (auto& Wi : r.W)
            Wi = Wi * c;
        return r;
    }
    VIOAlgebra operator-() const { return (*this) * -1.0; }
    VIOAlgebra operator+(const VIOAlgebra& o) const { // :168-186
        VIOAlgebra r = *this;
        r.u_beta = u_beta + o.u_beta;
        r.U_A = U_A + o.U_A;
        r.U_B = U_B + o.U_B;
        r.u_w = u_w + o.u_w;
        assert(W.size() == o.W.size());
        for (size_t i = 0; i < W.size(); ++i)
            r.W[i] = W[i] + o.W[i];
        return r;
    }
    VIOAlgebra operator-(const VIOAlgebra& o) const { return *this + (-o); }
};
inline VIOAlgebra operator*(double c, const VIOAlgebra& l) { return l * c; }

// VIOGroup.cpp:25-32
inline VIOSensorState sensorStateGroupAction(const VIOGroup& X, const VIOSensorState& s) {
    VIOSensorState r;
    r.inputBias = s.inputBias + X.beta;
    r.pose = s.pose * X.A;
    r.velocity = X.A.R.inverse() * (s.velocity - X.w);
    r.cameraOffset = X.A.inverse() * s.cameraOffset * X.B;
    return r;
}
// VIOGroup.cpp:34-55
inline VIOState stateGroupAction(const VIOGroup& X, const VIOState& state) {
    VIOState ns;
    ns.sensor = sensorStateGroupAction(X, state.sensor);
    assert(X.Q.size() == state.cameraLandmarks.size());
    ns.cameraLandmarks.resize(state.cameraLandmarks.size());
    for (size_t i = 0; i < X.Q.size(); ++i) {
        ns.cameraLandmarks[i].p = X.Q[i].inverse() * state.cameraLandmarks[i].p;
        ns.cameraLandmarks[i].id = state.cameraLandmarks[i].id;
    }
    return ns;
}
// VIOGroup.cpp:57-69
inline VisionMeasurement outputGroupAction(const VIOGroup& X, const VisionMeasurement& m) {
    VisionMeasurement r;
    for (size_t i = 0; i < X.Q.size(); ++i) {
        const auto it = m.camCoordinates.find(X.id[i]);
        if (it != m.camCoordinates.end()) {
            const Vec3 bearing = m.cameraPtr->undistortPoint(it->second);
            r.camCoordinates[X.id[i]] = m.cameraPtr->projectPoint(X.Q[i].R.inverse() * bearing);
        }
    }
    r.cameraPtr = m.cameraPtr;
    return r;
}
// VIOGroup.cpp:190-227
inline VIOAlgebra liftVelocity(const VIOState& state, const IMUVelocity& velocity) {
    VIOAlgebra lift;
    const VIOSensorState& s = state.sensor;
    const IMUVelocity v_est = velocity.minusBias(s.inputBias);
    lift.u_beta.setBlock<3, 1>(0, 0, velocity.gyrBiasVel);
    lift.u_beta.setBlock<3, 1>(3, 0, velocity.accBiasVel);
    lift.U_A.setBlock<3, 1>(0, 0, v_est.gyr);
    lift.U_A.setBlock<3, 1>(3, 0, s.velocity);
    lift.U_B = s.cameraOffset.inverse().Adjoint() * lift.U_A;
    lift.u_w = -v_est.acc + s.gravityDir() * GRAVITY_CONSTANT;
    const Vec6 U_C = s.cameraOffset.inverse().Adjoint() * lift.U_A;
    const Vec3 omega_C = U_C.block<3, 1>(0, 0);
    const Vec3 v_C = U_C.block<3, 1>(3, 0);
    lift.W.resize(state.cameraLandmarks.size());
    lift.id.resize(state.cameraLandmarks.size());
    for (size_t i = 0; i < state.cameraLandmarks.size(); ++i) {
        const Vec3& p = state.cameraLandmarks[i].p;
        Vec4 Wi;
        Wi.setBlock<3, 1>(0, 0, omega_C + skew(p) * v_C / p.squaredNorm());
        Wi(3) = dot(p, v_C) / p.squaredNorm();
        lift.W[i] = Wi;
        lift.id[i] = state.cameraLandmarks[i].id;
    }
    return lift;
}
// VIOGroup.cpp:229-271
inline VIOGroup liftVelocityDiscrete(const VIOState& state, const IMUVelocity& velocity, double dt) {
    VIOGroup lift;
    const VIOSensorState& s = state.sensor;
    const IMUVelocity v_est = velocity.minusBias(s.inputBias);
    lift.beta.setBlock<3, 1>(0, 0, dt * velocity.gyrBiasVel);
    lift.beta.setBlock<3, 1>(3, 0, dt * velocity.accBiasVel);
    lift.A.R = SO3::exp(dt * v_est.gyr);
    lift.A.x = dt * (s.pose.R * s.velocity) + 0.5 * dt * dt * (s.pose.R * v_est.acc + vec3(0, 0, -GRAVITY_CONSTANT));
    lift.A.x = s.pose.R.inverse() * lift.A.x;
    lift.B = s.cameraOffset.inverse() * lift.A * s.cameraOffset;
    const Vec3 bodyVelocityDiff = v_est.acc - s.gravityDir() * GRAVITY_CONSTANT;
    lift.w = s.velocity - (s.velocity + dt * bodyVelocityDiff);
    const int N = (int)state.cameraLandmarks.size();
    const SE3 cameraPoseChangeInv = s.cameraOffset.inverse() * lift.A.inverse() * s.cameraOffset;
    lift.Q.resize(N);
    lift.id.resize(N);
    for (int i = 0; i < N; ++i) {
        const Landmark& blm0 = state.cameraLandmarks[i];
        const Vec3 p1 = cameraPoseChangeInv * blm0.p;
        lift.Q[i].R = SO3::FromVectors(p1.normalized(), blm0.p.normalized());
        lift.Q[i].a = blm0.p.norm() / p1.norm();
        lift.id[i] = blm0.id;
    }
    return lift;
}
// VIOGroup.cpp:273-290
inline VIOGroup VIOExp(const VIOAlgebra& lambda) {
    M<9, 1> ext;
    ext.setBlock<6, 1>(0, 0, lambda.U_A);
    ext.setBlock<3, 1>(6, 0, lambda.u_w);
    const SE23 e = SE23::exp(ext);
    VIOGroup r;
    r.beta = lambda.u_beta;
    r.A = SE3(e.R, e.x0);
    r.w = e.x1;
    r.B = SE3::exp(lambda.U_B);
    r.id = lambda.id;
    r.Q.resize(lambda.W.size());
    for (size_t i = 0; i < lambda.W.size(); ++i)
        r.Q[i] = SOT3::exp(lambda.W[i]);
    return r;
}

// ---------------------------------------------------------------- numerical differential (src/mathematical/Geometry.cpp:25-36)
inline DMat numericalDifferential(const std::function<DVec(const DVec&)>& f, const DVec& x, double h = -1.0) {
    if (h < 0)
        h = std::cbrt(std::numeric_limits<double>::epsilon());
    const int rows = (int)f(x).size();
    const int cols = (int)x.size();
    DMat Df(rows, cols);
    for (int j = 0; j < cols; ++j) {
        DVec xp = x, xm = x;
        xp[j] += h;
        xm[j] -= h;
        const DVec fp = f(xp), fm = f(xm);
        for (int i = 0; i < rows; ++i)
            Df(i, j) = (fp[i] - fm[i]) / (2 * h);
    }
    return Df;
}

// ---------------------------------------------------------------- sphere charts (src/mathematical/VIOState.cpp:246-353)
inline Vec2 e3ProjectSphere(const Vec3& eta) { // :246-251
    const double s = 1.0 / (1.0 - eta(2));
    return vec2(eta(0) * s, eta(1) * s);
}
inline Vec3 e3ProjectSphereInv(const Vec2& y) { // :253-258
    const double k = 2.0 / (y.squaredNorm() + 1.0);
    return vec3(k * y(0), k * y(1), 1.0 + k * (0.0 - 1.0));
}
inline M<2, 3> e3ProjectSphereDiff(const Vec3& eta) { // :260-267
    const Vec3 e3 = vec3(0, 0, 1);
    const Mat3 inner = Mat3::Identity() * (1 - eta(2)) + (eta - e3) * e3.T();
    M<2, 3> D = inner.block<2, 3>(0, 0);
    return D * std::pow(1 - eta(2), -2.0);
}
inline M<3, 2> e3ProjectSphereInvDiff(const Vec2& y) { // :269-275
    M<3, 2> D;
    const M<2, 2> top = M<2, 2>::Identity() * (y.squaredNorm() + 1.0) - 2.0 * (y * y.T());
    D.setBlock<2, 2>(0, 0, top);
    D.setBlock<1, 2>(2, 0, 2.0 * y.T());
    return D * (2.0 * std::pow(y.squaredNorm() + 1.0, -2.0));
}
// stereographic chart about a pole (:282-307)
inline Vec2 sphereChart_stereo(const Vec3& eta, const Vec3& pole) {
    const SO3 rot = SO3::FromVectors(-pole, vec3(0, 0, 1));
    return e3ProjectSphere(rot * eta);
}
inline Vec3 sphereChart_stereo_inv(const Vec2& y, const Vec3& pole) {
    const Vec3 etaRot = e3ProjectSphereInv(y);
    const SO3 rot = SO3::FromVectors(-pole, vec3(0, 0, 1));
    return rot.inverse() * etaRot;
}
inline M<2, 3> sphereChart_stereo_diff0(const Vec3& pole) {
    const SO3 rot = SO3::FromVectors(-pole, vec3(0, 0, 1));
    const Vec3 etaRot = rot * pole;
    return e3ProjectSphereDiff(etaRot) * rot.asMatrix();
}
inline M<3, 2> sphereChart_stereo_inv_diff0(const Vec3& pole) {
    const SO3 rot = SO3::FromVectors(-pole, vec3(0, 0, 1));
    return rot.inverse().asMatrix() * e3ProjectSphereInvDiff(vec2(0, 0));
}
// normal chart about a pole (:309-353)
inline Vec2 sphereChart_normal(const Vec3& eta, const Vec3& pole) {
    const Vec3 e3 = vec3(0, 0, 1);
    const SO3 rot = SO3::FromVectors(pole, e3);
    const Vec3 y = rot * eta;
    const Vec3 yxe3 = skew(y) * e3;
    const double sin_th = yxe3.norm();
    const double cos_th = dot(y, e3);
    const double th = std::atan2(sin_th, cos_th);
    Vec3 omega;
    if (std::fabs(th) < 1e-8)
        omega = yxe3;
    else
        omega = yxe3 * (th / sin_th);
    return vec2(omega(0), omega(1));
}
inline Vec3 sphereChart_normal_inv(const Vec2& eps, const Vec3& pole) {
    const Vec3 e3 = vec3(0, 0, 1);
    const Vec3 omega = vec3(eps(0), eps(1), 0.0);
    const Vec3 y = SO3::exp(-omega) * e3;
    const SO3 rot = SO3::FromVectors(pole, e3);
    return rot.inverse() * y;
}
inline M<2, 3> sphereChart_normal_diff0(const Vec3& pole) {
    const SO3 rot = SO3::FromVectors(pole, vec3(0, 0, 1));
    M<2, 3> d = M<2, 3>::Zero();
    d(0, 1) = 1.0;
    d(1, 0) = -1.0;
    return d * rot.asMatrix();
}
inline M<3, 2> sphereChart_normal_inv_diff0(const Vec3& pole) {
    const SO3 rot = SO3::FromVectors(pole, vec3(0, 0, 1));
    M<3, 2> d = M<3, 2>::Zero();
    d(0, 1) = -1.0;
    d(1, 0) = 1.0;
    return rot.inverse().asMatrix() * d;
}

// ---------------------------------------------------------------- state charts (VIOState.cpp:104-244)
using M21 = M<21, 1>;
inline M21 sensorChart_std(const VIOSensorState& Xi, const VIOSensorState& Xi0) { // :104-121
    M21 eps;
    eps.setBlock<6, 1>(0, 0, Xi.inputBias - Xi0.inputBias);
    eps.setBlock<6, 1>(6, 0, SE3::log(Xi0.pose.inverse() * Xi.pose));
    eps.setBlock<3, 1>(12, 0, Xi.velocity - Xi0.velocity);
    eps.setBlock<6, 1>(15, 0, SE3::log(Xi0.cameraOffset.inverse() * Xi.cameraOffset));
    return eps;
}
inline VIOSensorState sensorChart_std_inv(const M21& eps, const VIOSensorState& Xi0) {
    VIOSensorState Xi;
    Xi.inputBias = Xi0.inputBias + eps.block<6, 1>(0, 0);
    Xi.pose = Xi0.pose * SE3::exp(eps.block<6, 1>(6, 0));
    Xi.velocity = Xi0.velocity + eps.block<3, 1>(12, 0);
    Xi.cameraOffset = Xi0.cameraOffset * SE3::exp(eps.block<6, 1>(15, 0));
    return Xi;
}
inline M21 sensorChart_normal(const VIOSensorState& Xi, const VIOSensorState& Xi0) { // :123-151
    const SE3 A = Xi0.pose.inverse() * Xi.pose;
    const Vec3 v_xi0 = Xi0.pose.R * Xi0.velocity;
    const Vec3 v_xi = Xi.pose.R * Xi.velocity;
    const Vec3 v_A = Xi0.pose.R.inverse() * (v_xi - v_xi0);
    const SE3 B = Xi0.cameraOffset.inverse() * A * Xi.cameraOffset;
    M21 eps;
    eps.setBlock<6, 1>(0, 0, Xi.inputBias - Xi0.inputBias);
    SE23 e;
    e.R = A.R;
    e.x0 = A.x;
    e.x1 = v_A;
    eps.setBlock<9, 1>(6, 0, SE23::log(e));
    eps.setBlock<6, 1>(15, 0, SE3::log(B));
    return eps;
}
inline VIOSensorState sensorChart_normal_inv(const M21& eps, const VIOSensorState& Xi0) {
    const SE23 X = SE23::exp(eps.block<9, 1>(6, 0));
    const SE3 B = SE3::exp(eps.block<6, 1>(15, 0));
    const SE3 A(X.R, X.x0);
    const Vec3 v_A = X.x1;
    VIOSensorState Xi;
    Xi.inputBias = Xi0.inputBias + eps.block<6, 1>(0, 0);
    Xi.pose = Xi0.pose * A;
    const Vec3 v_xi0 = Xi0.pose.R * Xi0.velocity;
    Xi.velocity = Xi.pose.R.inverse() * (v_xi0 + Xi0.pose.R * v_A);
    Xi.cameraOffset = A.inverse() * Xi0.cameraOffset * B;
    return Xi;
}
inline Vec3 pointChart_euclid(const Landmark& q, const Landmark& q0) { return q.p - q0.p; } // :153-157
inline Landmark pointChart_euclid_inv(const Vec3& eps, const Landmark& q0) { return Landmark{q0.p + eps, q0.id}; }
inline Vec3 pointChart_invdepth(const Landmark& q, const Landmark& q0) { // :159-186
    const double rho = 1.0 / q.p.norm();
    const double rho0 = 1.0 / q0.p.norm();
    const Vec3 y = q.p * rho;
    const Vec3 y0 = q0.p * rho0;
    const Vec2 s = sphereChart_stereo(y, y0);
    return vec3(s(0), s(1), rho - rho0);
}
inline Landmark pointChart_invdepth_inv(const Vec3& eps, const Landmark& q0) {
    const double rho0 = 1.0 / q0.p.norm();
    const Vec3 y0 = q0.p * rho0;
    const Vec3 y = sphereChart_stereo_inv(vec2(eps(0), eps(1)), y0);
    double rho = eps(2) + rho0;
    if (rho <= 0.0)
        rho = 1e-6;
    return Landmark{y / rho, q0.id};
}
inline Vec3 pointChart_normal(const Landmark& q, const Landmark& q0) { // :188-213
    const double rho = 1.0 / q.p.norm();
    const double rho0 = 1.0 / q0.p.norm();
    const Vec3 y = q.p * rho;
    const Vec3 y0 = q0.p * rho0;
    const Vec2 s = sphereChart_normal(y, y0);
    return vec3(s(0), s(1), std::log(rho / rho0));
}
inline Landmark pointChart_normal_inv(const Vec3& eps, const Landmark& q0) {
    const double rho0 = 1.0 / q0.p.norm();
    const Vec3 y0 = q0.p * rho0;
    const Vec3 y = sphereChart_normal_inv(vec2(eps(0), eps(1)), y0);
    const double rho = rho0 * std::exp(eps(2));
    return Landmark{y / rho, q0.id};
}

enum class CoordinateChoice { Euclidean = 0, InvDepth = 1, Normal = 2 };

// constructVIOChart (VIOState.cpp:215-244)
inline DVec VIOChart(CoordinateChoice cc, const VIOState& Xi, const VIOState& Xi0) {
    const size_t N = Xi.cameraLandmarks.size();
    assert(N == Xi0.cameraLandmarks.size());
    DVec eps(21 + 3 * N);
    const M21 s = (cc == CoordinateChoice::Normal) ? sensorChart_normal(Xi.sensor, Xi0.sensor)
                                                   : sensorChart_std(Xi.sensor, Xi0.sensor);
    for (int i = 0; i < 21; ++i)
        eps[i] = s(i);
    for (size_t i = 0; i < N; ++i) {
        Vec3 e;
        if (cc == CoordinateChoice::Euclidean)
            e = pointChart_euclid(Xi.cameraLandmarks[i], Xi0.cameraLandmarks[i]);
        else if (cc == CoordinateChoice::InvDepth)
            e = pointChart_invdepth(Xi.cameraLandmarks[i], Xi0.cameraLandmarks[i]);
        else
            e = pointChart_normal(Xi.cameraLandmarks[i], Xi0.cameraLandmarks[i]);
        for (int k = 0; k < 3; ++k)
            eps[21 + 3 * i + k] = e(k);
    }
    return eps;
}
inline VIOState VIOChartInv(CoordinateChoice cc, const DVec& eps, const VIOState& Xi0) {
    const size_t N = Xi0.cameraLandmarks.size();
    assert(eps.size() == 21 + 3 * N);
    VIOState Xi;
    M21 s;
    for (int i = 0; i < 21; ++i)
        s(i) = eps[i];
    Xi.sensor = (cc == CoordinateChoice::Normal) ? sensorChart_normal_inv(s, Xi0.sensor)
                                                 : sensorChart_std_inv(s, Xi0.sensor);
    Xi.cameraLandmarks.resize(N);
    for (size_t i = 0; i < N; ++i) {
        const Vec3 e = vec3(eps[21 + 3 * i], eps[21 + 3 * i + 1], eps[21 + 3 * i + 2]);
        if (cc == CoordinateChoice::Euclidean)
            Xi.cameraLandmarks[i] = pointChart_euclid_inv(e, Xi0.cameraLandmarks[i]);
        else if (cc == CoordinateChoice::InvDepth)
            Xi.cameraLandmarks[i] = pointChart_invdepth_inv(e, Xi0.cameraLandmarks[i]);
        else
            Xi.cameraLandmarks[i] = pointChart_normal_inv(e, Xi0.cameraLandmarks[i]);
    }
    return Xi;
}

// conv_euc2ind / conv_ind2euc (coordinateSuite/invdepth.cpp:65-81); same matrix as the diagonal block of
// coordinateDifferential_invdepth_euclid (VIOState.cpp:355-389)
inline Mat3 conv_euc2ind(const Vec3& q0) {
    const double rho0 = 1.0 / q0.norm();
    const Vec3 y0 = q0 * rho0;
    Mat3 Mc;
    const M<2, 3> top = rho0 * (sphereChart_stereo_diff0(y0) * (Mat3::Identity() - y0 * y0.T()));
    Mc.setBlock<2, 3>(0, 0, top);
    Mc.setBlock<1, 3>(2, 0, (-rho0 * rho0) * y0.T());
    return Mc;
}
inline Mat3 conv_ind2euc(const Vec3& q0) {
    const double rho0 = 1.0 / q0.norm();
    const Vec3 y0 = q0 * rho0;
    Mat3 Mc;
    Mc.setBlock<3, 2>(0, 0, sphereChart_stereo_inv_diff0(y0) / rho0);
    Mc.setBlock<3, 1>(0, 2, -y0 / (rho0 * rho0));
    return Mc;
}
// the ind2euc form used by liftInnovation_invdepth / C*_invdepth (invdepth.cpp:201-207, 257-262)
inline Mat3 ind2euc_r0(const Vec3& q0) {
    const double r0 = q0.norm();
    const Vec3 y0 = q0 / r0;
    Mat3 Mc;
    Mc.setBlock<3, 2>(0, 0, r0 * sphereChart_stereo_inv_diff0(y0));
    Mc.setBlock<3, 1>(0, 2, -r0 * q0);
    return Mc;
}
inline DMat coordinateDifferential_invdepth_euclid(const VIOState& Xi0) { // VIOState.cpp:355-389
    const int N = (int)Xi0.cameraLandmarks.size();
    DMat Mm = DMat::Identity(21 + 3 * N, 21 + 3 * N);
    for (int i = 0; i < N; ++i)
        Mm.setBlock<3, 3>(21 + 3 * i, 21 + 3 * i, conv_euc2ind(Xi0.cameraLandmarks[i].p));
    return Mm;
}
inline DMat coordinateDifferential_normal_euclid(const VIOState& Xi0) { // VIOState.cpp:391-401
    auto coordChange = [&](const DVec& eps) {
        return VIOChart(CoordinateChoice::Normal, VIOChartInv(CoordinateChoice::Euclidean, eps, Xi0), Xi0);
    };
    return numericalDifferential(coordChange, DVec(Xi0.Dim(), 0.0));
}

// ---------------------------------------------------------------- EqF matrices, Euclidean suite (coordinateSuite/euclid.cpp)
inline DMat EqFInputMatrixB_euclid(const VIOGroup& X, const VIOState& xi0) { // euclid.cpp:186-233
    const int N = (int)xi0.cameraLandmarks.size();
    DMat Bt(xi0.Dim(), 12);
    const VIOState xi_hat = stateGroupAction(X, xi0);
    Bt.setBlock<6, 6>(0, 6, Mat6::Identity());
    const Mat3 R_A = X.A.R.asMatrix();
    Bt.setBlock<3, 3>(6, 0, R_A);
    Bt.setBlock<3, 3>(9, 0, skew(X.A.x) * R_A);
    Bt.setBlock<3, 3>(12, 0, R_A * skew(xi_hat.sensor.velocity));
    Bt.setBlock<3, 3>(12, 3, R_A);
    const Mat3 RT_IC = xi_hat.sensor.cameraOffset.R.inverse().asMatrix();
    const Vec3 x_IC = xi_hat.sensor.cameraOffset.x;
    for (int i = 0; i < N; ++i) {
        const Mat3 Qhat_i = X.Q[i].R.asMatrix() * X.Q[i].a;
        const Vec3& qhat_i = xi_hat.cameraLandmarks[i].p;
        Bt.setBlock<3, 3>(21 + 3 * i, 0, Qhat_i * (skew(qhat_i) * RT_IC + RT_IC * skew(x_IC)));
    }
    return Bt;
}
inline DMat EqFStateMatrixA_euclid(const VIOGroup& X, const VIOState& xi0, const IMUVelocity& imuVel) { // euclid.cpp:99-160
    const int N = (int)xi0.cameraLandmarks.size();
    const int n = xi0.Dim();
    DMat A0t(n, n);
    const DMat Bt = EqFInputMatrixB_euclid(X, xi0);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < 6; ++j)
            A0t(i, j) = -Bt(i, j);
    A0t.setBlock<3, 3>(9, 12, Mat3::Identity());
    A0t.setBlock<3, 3>(12, 6, -GRAVITY_CONSTANT * skew(xi0.sensor.gravityDir()));
    const VIOState xi_hat = stateGroupAction(X, xi0);
    const IMUVelocity v_est = imuVel.minusBias(xi_hat.sensor.inputBias);
    Vec6 U_I;
    U_I.setBlock<3, 1>(0, 0, v_est.gyr);
    U_I.setBlock<3, 1>(3, 0, xi_hat.sensor.velocity);
    const Mat6 adTerm = SE3::adjoint(xi0.sensor.cameraOffset.inverse().Adjoint() * (X.A.Adjoint() * U_I));
    A0t.setBlock<6, 6>(15, 15, adTerm);
    const Mat3 R_IC = xi_hat.sensor.cameraOffset.R.asMatrix();
    const Mat3 R_Ahat = X.A.R.asMatrix();
    for (int i = 0; i < N; ++i) {
        const Mat3 Qhat_i = X.Q[i].R.asMatrix() * X.Q[i].a;
        A0t.setBlock<3, 3>(21 + 3 * i, 12, -(Qhat_i * R_IC.T() * R_Ahat.T()));
    }
    const Mat6 commonTerm = X.B.inverse().Adjoint() * adTerm;
    for (int i = 0; i < N; ++i) {
        M<3, 6> temp;
        temp.setBlock<3, 3>(0, 0, skew(xi0.cameraLandmarks[i].p) * X.Q[i].R.asMatrix());
        temp.setBlock<3, 3>(0, 3, -X.Q[i].a * X.Q[i].R.asMatrix());
        A0t.setBlock<3, 6>(21 + 3 * i, 15, temp * commonTerm);
    }
    const Vec6 U_C = xi_hat.sensor.cameraOffset.inverse().Adjoint() * U_I;
    const Vec3 v_C = U_C.block<3, 1>(3, 0);
    for (int i = 0; i < N; ++i) {
        const Mat3 Qhat_i = X.Q[i].R.asMatrix() * X.Q[i].a;
        const Vec3& qhat_i = xi_hat.cameraLandmarks[i].p;
        const Mat3 inner = skew(qhat_i) * skew(v_C) - 2.0 * (v_C * qhat_i.T()) + qhat_i * v_C.T();
        const Mat3 A_qi = -(Qhat_i * inner * inverse3(Qhat_i)) * (1.0 / qhat_i.squaredNorm());
        A0t.setBlock<3, 3>(21 + 3 * i, 21 + 3 * i, A_qi);
    }
    return A0t;
}
inline M<2, 3> EqFoutputMatrixCiStar_euclid(const Vec3& q0, const SOT3& QHat, const CameraPtr& cam, const Vec2& y) { // euclid.cpp:162-184
    const Vec3 qHat = QHat.inverse() * q0;
    const Vec3 yHat = qHat.normalized();
    M<4, 3> m2g;
    m2g.setBlock<3, 3>(0, 0, -skew(q0));
    m2g.setBlock<1, 3>(3, 0, -q0.T());
    m2g = m2g / q0.squaredNorm();
    auto DRho = [&cam](const Vec3& yVec) {
        M<3, 4> DRhoVec = M<3, 4>::Zero();
        DRhoVec.setBlock<3, 3>(0, 0, skew(yVec));
        return cam->projectionJacobian(yVec) * DRhoVec;
    };
    const Vec3 yTru = cam->undistortPoint(y);
    return (0.5 * (DRho(yTru) + DRho(yHat))) * QHat.inverse().Adjoint() * m2g;
}
inline VIOAlgebra liftInnovation_euclid(const DVec& g, const VIOState& xi0) { // euclid.cpp:36-69
    assert((int)g.size() == xi0.Dim());
    VIOAlgebra Delta;
    for (int i = 0; i < 6; ++i) {
        Delta.u_beta(i) = g[i];
        Delta.U_A(i) = g[6 + i];
    }
    const Vec3 gamma_v = vec3(g[12], g[13], g[14]);
    Delta.u_w = -gamma_v - skew(Delta.U_A.block<3, 1>(0, 0)) * xi0.sensor.velocity;
    Vec6 g15;
    for (int i = 0; i < 6; ++i)
        g15(i) = g[15 + i];
    Delta.U_B = g15 + xi0.sensor.cameraOffset.inverse().Adjoint() * Delta.U_A;
    const int N = (int)xi0.cameraLandmarks.size();
    Delta.id.resize(N);
    Delta.W.resize(N);
    for (int i = 0; i < N; ++i) {
        const Vec3 gq = vec3(g[21 + 3 * i], g[21 + 3 * i + 1], g[21 + 3 * i + 2]);
        const Vec3& qi0 = xi0.cameraLandmarks[i].p;
        Delta.W[i].setBlock<3, 1>(0, 0, -cross(qi0, gq) / qi0.squaredNorm());
        Delta.W[i](3) = -dot(qi0, gq) / qi0.squaredNorm();
        Delta.id[i] = xi0.cameraLandmarks[i].id;
    }
    return Delta;
}
inline VIOGroup liftInnovationDiscrete_sensor(const DVec& g, const VIOState& xi0) { // euclid.cpp:74-79 == invdepth.cpp:228-233
    VIOGroup lift;
    Vec6 g6, g15;
    for (int i = 0; i < 6; ++i) {
        lift.beta(i) = g[i];
        g6(i) = g[6 + i];
        g15(i) = g[15 + i];
    }
    lift.A = SE3::exp(g6);
    lift.w = xi0.sensor.velocity - lift.A.R * (xi0.sensor.velocity + vec3(g[12], g[13], g[14]));
    lift.B = xi0.sensor.cameraOffset.inverse() * lift.A * xi0.sensor.cameraOffset * SE3::exp(g15);
    return lift;
}
inline VIOGroup liftInnovationDiscrete_euclid(const DVec& g, const VIOState& xi0) { // euclid.cpp:71-97
    VIOGroup lift = liftInnovationDiscrete_sensor(g, xi0);
    const int N = (int)xi0.cameraLandmarks.size();
    lift.id.resize(N);
    lift.Q.resize(N);
    for (int i = 0; i < N; ++i) {
        const Vec3& qi = xi0.cameraLandmarks[i].p;
        const Vec3 qi1 = qi + vec3(g[21 + 3 * i], g[21 + 3 * i + 1], g[21 + 3 * i + 2]);
        lift.Q[i].R = SO3::FromVectors(qi1.normalized(), qi.normalized());
        lift.Q[i].a = qi.norm() / qi1.norm();
        lift.id[i] = xi0.cameraLandmarks[i].id;
    }
    return lift;
}

// ---------------------------------------------------------------- inverse-depth suite (coordinateSuite/invdepth.cpp)
inline DMat EqFInputMatrixB_invdepth(const VIOGroup& X, const VIOState& xi0) { // invdepth.cpp:123-181
    DMat Bt = EqFInputMatrixB_euclid(X, xi0); // sensor rows are identical (:152-166)
    const int N = (int)xi0.cameraLandmarks.size();
    for (int i = 0; i < N; ++i) {
        const Mat3 blk = conv_euc2ind(xi0.cameraLandmarks[i].p) * Bt.block<3, 3>(21 + 3 * i, 0);
        Bt.setBlock<3, 3>(21 + 3 * i, 0, blk);
    }
    return Bt;
}
inline DMat EqFStateMatrixA_invdepth(const VIOGroup& X, const VIOState& xi0, const IMUVelocity& imuVel) { // invdepth.cpp:36-121
    DMat A0t = EqFStateMatrixA_euclid(X, xi0, imuVel); // sensor block identical (:49-63)
    const int N = (int)xi0.cameraLandmarks.size();
    for (int i = 0; i < N; ++i) {
        const Vec3& q0 = xi0.cameraLandmarks[i].p;
        const Mat3 e2i = conv_euc2ind(q0);
        const int r = 21 + 3 * i;
        A0t.setBlock<3, 3>(r, 0, e2i * A0t.block<3, 3>(r, 0));   // -B[:,0:3] landmark rows (:47, :171-176)
        A0t.setBlock<3, 3>(r, 12, e2i * A0t.block<3, 3>(r, 12)); // :86-92
        A0t.setBlock<3, 6>(r, 15, e2i * A0t.block<3, 6>(r, 15)); // :97-103
        A0t.setBlock<3, 3>(r, r, e2i * A0t.block<3, 3>(r, r) * conv_ind2euc(q0)); // :108-117
    }
    return A0t;
}
inline M<2, 3> EqFoutputMatrixCiStar_invdepth(const Vec3& q0, const SOT3& QHat, const CameraPtr& cam, const Vec2& y) { // invdepth.cpp:255-266
    return EqFoutputMatrixCiStar_euclid(q0, QHat, cam, y) * ind2euc_r0(q0);
}
inline VIOAlgebra liftInnovation_invdepth(const DVec& g, const VIOState& xi0) { // invdepth.cpp:183-223
    DVec ge = g;
    const int N = (int)xi0.cameraLandmarks.size();
    for (int i = 0; i < N; ++i) {
        const Vec3 gq = ind2euc_r0(xi0.cameraLandmarks[i].p) * vec3(g[21 + 3 * i], g[21 + 3 * i + 1], g[21 + 3 * i + 2]);
        for (int k = 0; k < 3; ++k)
            ge[21 + 3 * i + k] = gq(k);
    }
    return liftInnovation_euclid(ge, xi0);
}
inline VIOGroup liftInnovationDiscrete_invdepth(const DVec& g, const VIOState& xi0) { // invdepth.cpp:225-253
    VIOGroup lift = liftInnovationDiscrete_sensor(g, xi0);
    const int N = (int)xi0.cameraLandmarks.size();
    lift.id.resize(N);
    lift.Q.resize(N);
    for (int i = 0; i < N; ++i) {
        const Landmark& q0i = xi0.cameraLandmarks[i];
        const Landmark q1i = pointChart_invdepth_inv(vec3(g[21 + 3 * i], g[21 + 3 * i + 1], g[21 + 3 * i + 2]), q0i);
        lift.Q[i].R = SO3::FromVectors(q1i.p.normalized(), q0i.p.normalized());
        lift.Q[i].a = q0i.p.norm() / q1i.p.norm();
        lift.id[i] = q0i.id;
    }
    return lift;
}

// ---------------------------------------------------------------- normal suite (coordinateSuite/normal.cpp:37-65)
inline M<2, 3> EqFoutputMatrixCiStar_normal(const Vec3& q0, const SOT3& QHat, const CameraPtr& cam, const Vec2&) {
    const Vec3 y0 = q0.normalized();
    const Vec3 yHat = QHat.R.inverse() * y0;
    M<2, 3> C0i = M<2, 3>::Zero();
    const M<2, 2> blk = cam->projectionJacobian(yHat) * QHat.R.asMatrix().T() * sphereChart_normal_inv_diff0(q0);
    C0i.setBlock<2, 2>(0, 0, blk);
    return C0i;
}

// ---------------------------------------------------------------- suite dispatch (include/eqvio/mathematical/EqFMatrices.h:35-90)
struct EqFCoordinateSuite {
    CoordinateChoice cc;
    DVec stateChart(const VIOState& xi, const VIOState& xi0) const { return VIOChart(cc, xi, xi0); }
    VIOState stateChartInv(const DVec& eps, const VIOState& xi0) const { return VIOChartInv(cc, eps, xi0); }
    DMat stateMatrixA(const VIOGroup& X, const VIOState& xi0, const IMUVelocity& v) const {
        if (cc == CoordinateChoice::Euclidean)
            return EqFStateMatrixA_euclid(X, xi0, v);
        if (cc == CoordinateChoice::InvDepth)
            return EqFStateMatrixA_invdepth(X, xi0, v);
        const DMat Mm = coordinateDifferential_normal_euclid(xi0);
        return Mm * EqFStateMatrixA_euclid(X, xi0, v) * lu_inverse(Mm);
    }
    DMat inputMatrixB(const VIOGroup& X, const VIOState& xi0) const {
        if (cc == CoordinateChoice::Euclidean)
            return EqFInputMatrixB_euclid(X, xi0);
        if (cc == CoordinateChoice::InvDepth)
            return EqFInputMatrixB_invdepth(X, xi0);
        return coordinateDifferential_normal_euclid(xi0) * EqFInputMatrixB_euclid(X, xi0);
    }
    M<2, 3> outputMatrixCiStar(const Vec3& q0, const SOT3& QHat, const CameraPtr& cam, const Vec2& y) const {
        if (cc == CoordinateChoice::Euclidean)
            return EqFoutputMatrixCiStar_euclid(q0, QHat, cam, y);
        if (cc == CoordinateChoice::InvDepth)
            return EqFoutputMatrixCiStar_invdepth(q0, QHat, cam, y);
        return EqFoutputMatrixCiStar_normal(q0, QHat, cam, y);
    }
    M<2, 3> outputMatrixCi(const Vec3& q0, const SOT3& QHat, const CameraPtr& cam) const { // EqFMatrices.cpp:84-89
        const Vec3 qHat = QHat.inverse() * q0;
        return outputMatrixCiStar(q0, QHat, cam, cam->projectPoint(qHat));
    }
    VIOAlgebra liftInnovation(const DVec& g, const VIOState& xi0) const {
        if (cc == CoordinateChoice::Euclidean)
            return liftInnovation_euclid(g, xi0);
        if (cc == CoordinateChoice::InvDepth)
            return liftInnovation_invdepth(g, xi0);
        const DMat Mi = lu_inverse(coordinateDifferential_normal_euclid(xi0));
        return liftInnovation_euclid(matvec(Mi, g), xi0);
    }
    VIOGroup liftInnovationDiscrete(const DVec& g, const VIOState& xi0) const {
        if (cc == CoordinateChoice::Euclidean)
            return liftInnovationDiscrete_euclid(g, xi0);
        if (cc == CoordinateChoice::InvDepth)
            return liftInnovationDiscrete_invdepth(g, xi0);
        return liftInnovationDiscrete_euclid(
            VIOChart(CoordinateChoice::Euclidean, VIOChartInv(CoordinateChoice::Normal, g, xi0), xi0), xi0);
    }
    // EqFMatrices.cpp:43-82. Rows follow ascending measurement id; columns follow the state's landmark order.
    DMat outputMatrixC(const VIOState& xi0, const VIOGroup& X, const VisionMeasurement& y, bool useEquivariance = true) const {
        const int Mn = (int)xi0.cameraLandmarks.size();
        const std::vector<int> ids = y.getIds();
        const int N = (int)ids.size();
        DMat CStar(2 * N, 21 + 3 * Mn);
        for (int i = 0; i < Mn; ++i) {
            const int idNum = xi0.cameraLandmarks[i].id;
            const Vec3& qi0 = xi0.cameraLandmarks[i].p;
            const auto it_y = std::find(ids.begin(), ids.end(), idNum);
            const auto it_Q = std::find(X.id.begin(), X.id.end(), idNum);
            assert(it_Q != X.id.end());
            const int k = (int)std::distance(X.id.begin(), it_Q);
            if (it_y != ids.end()) {
                const int j = (int)std::distance(ids.begin(), it_y);
                const M<2, 3> blk = useEquivariance ? outputMatrixCiStar(qi0, X.Q[k], y.cameraPtr, y.camCoordinates.at(idNum))
                                                    : outputMatrixCi(qi0, X.Q[k], y.cameraPtr);
                CStar.setBlock<2, 3>(2 * j, 21 + 3 * i, blk);
            }
        }
        return CStar;
    }
    // EqFMatrices.cpp:24-41
    DMat stateMatrixADiscrete(const VIOGroup& X, const VIOState& xi0, const IMUVelocity& imuVel, double dt) const {
        auto a0Discrete = [&](const DVec& epsilon) {
            const VIOState xi_e = stateChartInv(epsilon, xi0);
            const VIOState xi_hat = stateGroupAction(X, xi0);
            const VIOState xi = stateGroupAction(X, xi_e);
            const VIOGroup LambdaTilde = liftVelocityDiscrete(xi, imuVel, dt) * liftVelocityDiscrete(xi_hat, imuVel, dt).inverse();
            const VIOState xi_e1 = stateGroupAction(X * LambdaTilde * X.inverse(), xi_e);
            return stateChart(xi_e1, xi0);
        };
        return numericalDifferential(a0Discrete, DVec(xi0.Dim(), 0.0));
    }
};
inline const EqFCoordinateSuite* getCoordinates(CoordinateChoice cc) {
    static const EqFCoordinateSuite suites[3] = {{CoordinateChoice::Euclidean}, {CoordinateChoice::InvDepth}, {CoordinateChoice::Normal}};
    return &suites[(int)cc];
}

// ---------------------------------------------------------------- the EqF (src/mathematical/VIO_eqf.cpp)
inline void removeRowsCols(DMat& mat, int start, int num) { // VIO_eqf.cpp:27-45 (rows then cols)
    const int n = mat.r;
    DMat out(n - num, n - num);
    for (int j = 0, jo = 0; j < n; ++j) {
        if (j >= start && j < start + num)
            continue;
        for (int i = 0, io = 0; i < n; ++i) {
            if (i >= start && i < start + num)
                continue;
            out(io, jo) = mat(i, j);
            ++io;
        }
        ++jo;
    }
    mat = out;
}

enum class UpdateArithmetic {
    AsWritten = 0, // VIO_eqf.cpp:116-131 literally: LU inverse, K and S^-1 evaluated twice (lazy Eigen exprs), (K*C)*Sigma
    Reference = 1, // same formulas evaluated once (identical values to AsWritten); the parity oracle
    Efficient = 2  // dense T=Sigma C^T, S=C T+R, Cholesky, K by two triangular solves, Sigma -= K T^T (BASELINE.md §2 "U")
};

struct VIO_eqf {
    const EqFCoordinateSuite* coordinateSuite = getCoordinates(CoordinateChoice::Euclidean);
    VIOState xi0;
    VIOGroup X = VIOGroup::Identity();
    DMat Sigma = DMat::Identity(21, 21);
    double currentTime = -1;
    UpdateArithmetic arithmetic = UpdateArithmetic::Reference;
    DVec lastGamma; // kept for parity checks (Gamma = K yTilde of the last update)

    VIOState stateEstimate() const { return stateGroupAction(X, xi0); } // :137

    void integrateObserverState(const IMUVelocity& imu, double dt, bool discreteLift = true) { // :47-60
        VIOGroup lifted;
        if (discreteLift)
            lifted = liftVelocityDiscrete(stateEstimate(), imu, dt);
        else
            lifted = VIOExp(dt * liftVelocity(stateEstimate(), imu));
        X = X * lifted;
    }
    static DMat addDiag(DMat Mm, const DVec& diag, double s) {
        for (int i = 0; i < Mm.r; ++i)
            Mm(i, i) += s * diag[i];
        return Mm;
    }
    // Q (12x12) and P (nxn) are diagonal in the reference (VIOFilterSettings.h:176-201); passed as diagonals.
    void integrateRiccatiStateFast(const IMUVelocity& imu, double dt, const DVec& Qdiag, const DVec& Pdiag) { // :62-72
        const DMat A0t = coordinateSuite->stateMatrixA(X, xi0, imu);
        const DMat Bt = coordinateSuite->inputMatrixB(X, xi0);
        const int n = xi0.Dim();
        const DMat A0tExp = DMat::Identity(n, n) + dt * A0t;
        DMat BQ = Bt;
        for (int j = 0; j < 12; ++j)
            for (int i = 0; i < n; ++i)
                BQ(i, j) *= Qdiag[j];
        Sigma = A0tExp * Sigma * A0tExp.T() + dt * addDiag(BQ * Bt.T(), Pdiag, 1.0);
    }
    void integrateRiccatiStateAccurate(const IMUVelocity& imu, double dt, const DVec& Qdiag, const DVec& Pdiag) { // :74-91
        const DMat A0t = coordinateSuite->stateMatrixA(X, xi0, imu);
        const DMat Bt = coordinateSuite->inputMatrixB(X, xi0);
        const int n = xi0.Dim();
        DMat AB(n + 12, n + 12);
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i)
                AB(i, j) = A0t(i, j);
        for (int j = 0; j < 12; ++j)
            for (int i = 0; i < n; ++i)
                AB(i, n + j) = Bt(i, j);
        const DMat ABExp = expm(dt * AB);
        DMat A0tExp(n, n), BtExp(n, 12);
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i)
                A0tExp(i, j) = ABExp(i, j);
        for (int j = 0; j < 12; ++j)
            for (int i = 0; i < n; ++i)
                BtExp(i, j) = ABExp(i, n + j);
        DMat BQ = BtExp;
        for (int j = 0; j < 12; ++j)
            for (int i = 0; i < n; ++i)
                BQ(i, j) *= Qdiag[j] / dt;
        Sigma = addDiag(A0tExp * Sigma * A0tExp.T() + BQ * BtExp.T(), Pdiag, dt);
    }
    void integrateRiccatiStateDiscrete(const IMUVelocity& imu, double dt, const DVec& Qdiag, const DVec& Pdiag) { // :93-103
        const DMat Bt = coordinateSuite->inputMatrixB(X, xi0);
        const DMat Ad = coordinateSuite->stateMatrixADiscrete(X, xi0, imu, dt);
        const int n = xi0.Dim();
        DMat BQ = Bt;
        for (int j = 0; j < 12; ++j)
            for (int i = 0; i < n; ++i)
                BQ(i, j) *= Qdiag[j];
        Sigma = Ad * Sigma * Ad.T() + dt * addDiag(BQ * Bt.T(), Pdiag, 1.0);
    }

    // :105-135. R = measNoiseVar * I (VIOFilterSettings.h:203-206)
    void performVisionUpdate(const VisionMeasurement& measurement, double measNoiseVar, bool useEquivariantOutput = true, bool discreteCorrection = false) {
        if (measurement.camCoordinates.empty())
            return;
        const VisionMeasurement estimated = measureSystemState(stateEstimate(), measurement.cameraPtr);
        const DVec yTilde = (measurement - estimated).asVector();
        const DMat Ct = coordinateSuite->outputMatrixC(xi0, X, measurement, useEquivariantOutput);
        const int m = Ct.r;
        DVec Gamma;
        if (arithmetic == UpdateArithmetic::Efficient) {
            const DMat T = Sigma * Ct.T();
            DMat S = Ct * T;
            for (int i = 0; i < m; ++i)
                S(i, i) += measNoiseVar;
            DMat L;
            if (!cholesky_lower(S, L))
                throw std::runtime_error("oracle: S not positive definite");
            DMat K = T;
            trsm_right_lower_trans(L, K); // K = T L^-T
            trsm_right_lower(L, K);       // K = T L^-T L^-1 = T S^-1
            Gamma = matvec(K, yTilde);
            Sigma = Sigma - K * T.T();
        } else {
            auto SInvF = [&]() {
                DMat S = Ct * Sigma * Ct.T();
                for (int i = 0; i < m; ++i)
                    S(i, i) += measNoiseVar;
                return lu_inverse(S);
            };
            auto KF = [&]() { return Sigma * Ct.T() * SInvF(); };
            if (arithmetic == UpdateArithmetic::AsWritten) {
                Gamma = matvec(KF(), yTilde);
                Sigma = Sigma - KF() * Ct * Sigma;
            } else {
                const DMat K = KF();
                Gamma = matvec(K, yTilde);
                Sigma = Sigma - K * Ct * Sigma;
            }
        }
        lastGamma = Gamma;
        VIOGroup Delta;
        if (discreteCorrection)
            Delta = coordinateSuite->liftInnovationDiscrete(Gamma, xi0);
        else
            Delta = VIOExp(coordinateSuite->liftInnovation(Gamma, xi0));
        X = Delta * X;
    }

    double computeNEES(const VIOState& trueState) const { // :153-170
        VIOState truncated;
        truncated.sensor = trueState.sensor;
        for (const int id : X.id) {
            const auto it = std::find_if(trueState.cameraLandmarks.begin(), trueState.cameraLandmarks.end(), [&id](const Landmark& lm) { return lm.id == id; });
            assert(it != trueState.cameraLandmarks.end());
            truncated.cameraLandmarks.push_back(*it);
        }
        const VIOState stateError = stateGroupAction(X.inverse(), truncated);
        const DVec eps = coordinateSuite->stateChart(stateError, xi0);
        const DMat info = lu_inverse(Sigma);
        const DVec ie = matvec(info, eps);
        double nees = 0;
        for (size_t i = 0; i < eps.size(); ++i)
            nees += eps[i] * ie[i];
        return nees / truncated.Dim();
    }
    void removeLandmarkByIndex(int idx) { // :172-178
        xi0.cameraLandmarks.erase(xi0.cameraLandmarks.begin() + idx);
        X.id.erase(X.id.begin() + idx);
        X.Q.erase(X.Q.begin() + idx);
        removeRowsCols(Sigma, 21 + 3 * idx, 3);
    }
    void removeLandmarkById(int id) { // :180-186
        const auto it = std::find_if(xi0.cameraLandmarks.begin(), xi0.cameraLandmarks.end(), [&id](const Landmark& lm) { return lm.id == id; });
        assert(it != xi0.cameraLandmarks.end());
        removeLandmarkByIndex((int)std::distance(xi0.cameraLandmarks.begin(), it));
    }
    Mat3 getLandmarkCovById(int id) const { // :188-194
        const auto it = std::find_if(xi0.cameraLandmarks.begin(), xi0.cameraLandmarks.end(), [&id](const Landmark& lm) { return lm.id == id; });
        assert(it != xi0.cameraLandmarks.end());
        const int i = (int)std::distance(xi0.cameraLandmarks.begin(), it);
        return Sigma.block<3, 3>(21 + 3 * i, 21 + 3 * i);
    }
    M<2, 2> getOutputCovById(int id, const Vec2&, const CameraPtr& cam) const { // :196-211
        const Mat3 lmCov = getLandmarkCovById(id);
        const auto it = std::find_if(xi0.cameraLandmarks.begin(), xi0.cameraLandmarks.end(), [&id](const Landmark& lm) { return lm.id == id; });
        const auto it_X = std::find(X.id.begin(), X.id.end(), it->id);
        const SOT3& Q_i = X.Q[std::distance(X.id.begin(), it_X)];
        const M<2, 3> C0i = coordinateSuite->outputMatrixCi(it->p, Q_i, cam);
        return C0i * lmCov * C0i.T();
    }
    void removeInvalidLandmarks() { // :213-223
        std::set<int> invalid;
        for (size_t i = 0; i < X.id.size(); ++i)
            if (X.Q[i].a <= 1e-8 || X.Q[i].a > 1e8)
                invalid.insert(X.id[i]);
        for (const int id : invalid)
            removeLandmarkById(id);
    }
    // :225-245 ; newLandmarkCov is always a scaled identity at the call sites (VIOFilter.cpp:129-130, 274-276)
    void addNewLandmarks(const std::vector<Landmark>& newLandmarks, double newLandmarkVar) {
        xi0.cameraLandmarks.insert(xi0.cameraLandmarks.end(), newLandmarks.begin(), newLandmarks.end());
        for (const auto& lm : newLandmarks) {
            X.id.push_back(lm.id);
            X.Q.push_back(SOT3::Identity());
        }
        const int og = Sigma.r;
        const int newN = (int)newLandmarks.size();
        DMat S2(og + 3 * newN, og + 3 * newN);
        for (int j = 0; j < og; ++j)
            for (int i = 0; i < og; ++i)
                S2(i, j) = Sigma(i, j);
        for (int i = og; i < og + 3 * newN; ++i)
            S2(i, i) = newLandmarkVar;
        Sigma = S2;
    }
    VIOState predictState(double stamp, const std::vector<IMUVelocity>& imus) const { // :139-151
        VIOState pred = stateEstimate();
        for (size_t i = 0; i < imus.size(); ++i) {
            const double t0 = std::max(imus[i].stamp, currentTime);
            const double t1 = i + 1 < imus.size() ? std::min(imus[i + 1].stamp, stamp) : stamp;
            const double dt = std::max(t1 - t0, 0.0);
            pred = integrateSystemFunction(pred, imus[i], dt);
        }
        return pred;
    }
};

// ---------------------------------------------------------------- settings (include/eqvio/VIOFilterSettings.h:58-229)
struct Settings {
    double biasOmegaProcessVariance = 0.001, biasAccelProcessVariance = 0.001, attitudeProcessVariance = 0.001,
           positionProcessVariance = 0.001, velocityProcessVariance = 0.001, cameraAttitudeProcessVariance = 0.001,
           cameraPositionProcessVariance = 0.001, pointProcessVariance = 0.001;
    double velGyrNoise = 1e-4, velAccNoise = 1e-3, velGyrBiasWalk = 1e-5, velAccBiasWalk = 1e-3;
    double measurementNoise = 2.0, outlierThresholdAbs = 1e8, outlierThresholdProb = 1e8, featureRetention = 0.3;
    double initialAttitudeVariance = 1.0e-4, initialPositionVariance = 1.0e-4, initialVelocityVariance = 1.0e-2,
           initialCameraAttitudeVariance = 1.0e-5, initialCameraPositionVariance = 1.0e-4, initialPointVariance = 1.0,
           initialPointDepthVariance = -1.0, initialBiasOmegaVariance = 0.1, initialBiasAccelVariance = 0.1,
           initialSceneDepth = 1.0;
    bool useDiscreteInnovationLift = true, useDiscreteVelocityLift = true, useDiscreteStateMatrix = false,
         fastRiccati = false, useMedianDepth = true, useFeaturePredictions = false, useEquivariantOutput = true,
         removeLostLandmarks = true;
    CoordinateChoice coordinateChoice = CoordinateChoice::Euclidean;
    SE3 cameraOffset;

    DVec stateGainDiag(size_t N) const { // constructStateGainMatrix :176-190
        DVec P(21 + 3 * N, pointProcessVariance);
        const double v[7] = {biasOmegaProcessVariance, biasAccelProcessVariance, attitudeProcessVariance, positionProcessVariance,
                             velocityProcessVariance, cameraAttitudeProcessVariance, cameraPositionProcessVariance};
        for (int b = 0; b < 7; ++b)
            for (int k = 0; k < 3; ++k)
                P[3 * b + k] = v[b];
        return P;
    }
    DVec inputGainDiag() const { // constructInputGainMatrix :192-201
        DVec Q(12);
        const double v[4] = {velGyrNoise * velGyrNoise, velAccNoise * velAccNoise, velGyrBiasWalk * velGyrBiasWalk, velAccBiasWalk * velAccBiasWalk};
        for (int b = 0; b < 4; ++b)
            for (int k = 0; k < 3; ++k)
                Q[3 * b + k] = v[b];
        return Q;
    }
    double outputGainVar() const { return measurementNoise * measurementNoise; } // :203-206
    DVec initialCovDiag(size_t N = 0) const { // constructInitialStateCovariance :208-229
        DVec S(21 + 3 * N, initialPointVariance);
        const double v[7] = {initialBiasOmegaVariance, initialBiasAccelVariance, initialAttitudeVariance, initialPositionVariance,
                             initialVelocityVariance, initialCameraAttitudeVariance, initialCameraPositionVariance};
        for (int b = 0; b < 7; ++b)
            for (int k = 0; k < 3; ++k)
                S[3 * b + k] = v[b];
        if (initialPointDepthVariance > 0)
            for (size_t i = 0; i < N; ++i)
                S[21 + 3 * i + 2] = initialPointDepthVariance;
        return S;
    }
    DMat constructInitialStateCovariance(size_t N = 0) const {
        const DVec dgl = initialCovDiag(N);
        DMat S((int)dgl.size(), (int)dgl.size());
        for (size_t i = 0; i < dgl.size(); ++i)
            S((int)i, (int)i) = dgl[i];
        return S;
    }
};

// ---------------------------------------------------------------- the filter wrapper (src/VIOFilter.cpp)
class VIOFilter {
  public:
    VIO_eqf filterState;
    bool initialisedFlag = false;
    std::vector<IMUVelocity> velocityBuffer;
    Settings settings;

    VIOFilter() = default;
    explicit VIOFilter(const Settings& s) : settings(s) { // VIOFilter.cpp:31-41
        filterState.Sigma = settings.constructInitialStateCovariance();
        filterState.xi0.sensor.cameraOffset = s.cameraOffset;
        filterState.coordinateSuite = getCoordinates(settings.coordinateChoice);
    }
    VIOFilter(const VIOState& xi0, const Settings& s, double time = 0.0) : settings(s) { // :43-56
        filterState.Sigma = settings.constructInitialStateCovariance(xi0.cameraLandmarks.size());
        filterState.xi0 = xi0;
        for (const Landmark& lm : xi0.cameraLandmarks) {
            filterState.X.Q.push_back(SOT3::Identity());
            filterState.X.id.push_back(lm.id);
        }
        filterState.coordinateSuite = getCoordinates(s.coordinateChoice);
        filterState.currentTime = time;
        initialisedFlag = true;
    }
    void processIMUData(const IMUVelocity& imu) { // :58-63
        if (!initialisedFlag)
            initialiseFromIMUData(imu);
        velocityBuffer.push_back(imu);
    }
    void initialiseFromIMUData(const IMUVelocity& imu) { // :65-78
        filterState.xi0.sensor.inputBias = Vec6::Zero();
        filterState.xi0.sensor.pose = SE3::Identity();
        filterState.xi0.sensor.velocity = Vec3::Zero();
        initialisedFlag = true;
        filterState.xi0.sensor.pose.R = SO3::FromVectors(imu.acc.normalized(), vec3(0, 0, 1));
        filterState.currentTime = imu.stamp;
    }
    void setState(const VIOState& xi) { // :80-92
        filterState.xi0 = xi;
        filterState.X = VIOGroup::Identity(xi.getIds());
        const int N = (int)xi.cameraLandmarks.size();
        DMat S = DMat::Identity(21 + 3 * N, 21 + 3 * N);
        const DVec d0 = settings.initialCovDiag(0);
        for (int i = 0; i < 21; ++i)
            S(i, i) = d0[i];
        for (int i = 21; i < 21 + 3 * N; ++i)
            S(i, i) *= settings.initialPointVariance;
        filterState.Sigma = S;
        initialisedFlag = true;
    }
    void setLandmarks(const std::vector<Landmark>& lms) { // :94-110
        const DVec full = settings.initialCovDiag(lms.size());
        const int k = 3 * (int)lms.size();
        for (int j = 0; j < k; ++j)
            for (int i = 0; i < k; ++i)
                filterState.Sigma(21 + i, 21 + j) = (i == j) ? full[21 + i] : 0.0;
        filterState.xi0.cameraLandmarks = lms;
        filterState.X.Q.clear();
        filterState.X.id.clear();
        for (const Landmark& lm : lms) {
            filterState.X.Q.push_back(SOT3::Identity());
            filterState.X.id.push_back(lm.id);
        }
    }
    void augmentLandmarkStates(const std::vector<int>& newIds, const VIOState& provided) { // :112-132
        removeOldLandmarks(newIds);
        std::vector<Landmark> newLandmarks;
        for (const int id : newIds) {
            if (std::find(filterState.X.id.begin(), filterState.X.id.end(), id) != filterState.X.id.end())
                continue;
            const auto it2 = std::find_if(provided.cameraLandmarks.begin(), provided.cameraLandmarks.end(), [&id](const Landmark& lm) { return lm.id == id; });
            newLandmarks.push_back(*it2);
        }
        filterState.addNewLandmarks(newLandmarks, settings.initialPointVariance);
    }
    bool integrateUpToTime(double newTime) { // :134-192
        if (newTime <= filterState.currentTime || filterState.currentTime < 0 || velocityBuffer.empty())
            return false;
        auto interval = [&](size_t i) {
            const double t0 = std::max(velocityBuffer[i].stamp, filterState.currentTime);
            const double t1 = i + 1 < velocityBuffer.size() ? std::min(velocityBuffer[i + 1].stamp, newTime) : newTime;
            return std::max(t1 - t0, 0.0);
        };
        const size_t N = filterState.xi0.cameraLandmarks.size();
        if (settings.fastRiccati) {
            double accumulatedTime = 0;
            IMUVelocity acc = IMUVelocity::Zero();
            for (size_t i = 0; i < velocityBuffer.size(); ++i) {
                const double dt = interval(i);
                accumulatedTime += dt;
                acc = acc + velocityBuffer[i] * dt;
            }
            acc = acc * (1.0 / accumulatedTime);
            filterState.integrateRiccatiStateFast(acc, accumulatedTime, settings.inputGainDiag(), settings.stateGainDiag(N));
        }
        for (size_t i = 0; i < velocityBuffer.size(); ++i) {
            const double dt = interval(i);
            if (!settings.fastRiccati && dt > 0) {
                if (settings.useDiscreteStateMatrix)
                    filterState.integrateRiccatiStateDiscrete(velocityBuffer[i], dt, settings.inputGainDiag(), settings.stateGainDiag(N));
                else
                    filterState.integrateRiccatiStateAccurate(velocityBuffer[i], dt, settings.inputGainDiag(), settings.stateGainDiag(N));
            }
            filterState.integrateObserverState(velocityBuffer[i], dt, settings.useDiscreteVelocityLift);
        }
        filterState.currentTime = newTime;
        auto it = std::find_if(velocityBuffer.begin(), velocityBuffer.end(), [this](const IMUVelocity& v) { return v.stamp >= filterState.currentTime; });
        if (it != velocityBuffer.begin()) {
            --it;
            velocityBuffer.erase(velocityBuffer.begin(), it);
        }
        return true;
    }
    void processVisionData(const VisionMeasurement& measurement) { // :194-241
        const bool integrationFlag = integrateUpToTime(measurement.stamp);
        if (!integrationFlag || !initialisedFlag)
            return;
        if (settings.removeLostLandmarks)
            removeOldLandmarks(measurement.getIds());
        VisionMeasurement matched = measurement;
        removeOutliers(matched);
        addNewLandmarks(matched);
        if (matched.camCoordinates.empty())
            return;
        filterState.performVisionUpdate(matched, settings.outputGainVar(), settings.useEquivariantOutput, settings.useDiscreteInnovationLift);
        filterState.removeInvalidLandmarks();
    }
    VIOState stateEstimate() const { return filterState.stateEstimate(); }
    double getTime() const { return filterState.currentTime; }
    VisionMeasurement getFeaturePredictions(const CameraPtr& cam, double stamp) const { // :247-252
        if (settings.useFeaturePredictions)
            return measureSystemState(filterState.predictState(stamp, velocityBuffer), cam);
        return VisionMeasurement();
    }
    void addNewLandmarks(const VisionMeasurement& measurement) { // :258-278
        std::vector<Landmark> newLandmarks;
        for (const auto& cc : measurement.camCoordinates) {
            const int ccId = cc.first;
            if (std::none_of(filterState.X.id.begin(), filterState.X.id.end(), [&ccId](int i) { return i == ccId; }))
                newLandmarks.push_back(Landmark{measurement.cameraPtr->undistortPoint(cc.second), ccId});
        }
        if (newLandmarks.empty())
            return;
        const double initialDepth = settings.useMedianDepth ? getMedianSceneDepth() : settings.initialSceneDepth;
        for (auto& blm : newLandmarks)
            blm.p = blm.p * initialDepth;
        filterState.addNewLandmarks(newLandmarks, settings.initialPointVariance);
    }
    void removeOldLandmarks(const std::vector<int>& measurementIds) { // :280-302
        std::vector<int> lost;
        for (int i = 0; i < (int)filterState.X.id.size(); ++i) {
            const int oldId = filterState.X.id[i];
            if (std::find(measurementIds.begin(), measurementIds.end(), oldId) == measurementIds.end())
                lost.push_back(i);
        }
        std::reverse(lost.begin(), lost.end());
        for (const int li : lost)
            filterState.removeLandmarkByIndex(li);
    }
    void removeOutliers(VisionMeasurement& measurement) { // :304-364
        const size_t maxOutliers = (size_t)((1.0 - settings.featureRetention) * measurement.camCoordinates.size());
        const VIOState xiHat = stateEstimate();
        const VisionMeasurement yHat = measureSystemState(xiHat, measurement.cameraPtr);
        std::vector<int> proposed;
        std::map<int, double> absoluteOutliers;
        for (const auto& kv : yHat.camCoordinates) {
            const int lmId = kv.first;
            if (measurement.camCoordinates.count(lmId) == 0)
                continue;
            const double errAbs = (measurement.camCoordinates.at(lmId) - kv.second).norm();
            if (errAbs > settings.outlierThresholdAbs) {
                absoluteOutliers[lmId] = errAbs;
                proposed.push_back(lmId);
            }
        }
        std::map<int, double> probabilisticOutliers;
        const VisionMeasurement residual = measurement - yHat;
        for (const auto& kv : residual.camCoordinates) {
            const int lmId = kv.first;
            if (absoluteOutliers.count(lmId) || measurement.camCoordinates.count(lmId) == 0)
                continue;
            const M<2, 2> outputCov = filterState.getOutputCovById(lmId, measurement.camCoordinates[lmId], measurement.cameraPtr);
            const Vec2 t = inverse2(outputCov) * kv.second;
            const double errProb = dot(kv.second, t);
            if (errProb > settings.outlierThresholdProb) {
                probabilisticOutliers[lmId] = errProb;
                proposed.push_back(lmId);
            }
        }
        std::sort(proposed.begin(), proposed.end(), [&](int a, int b) {
            if (absoluteOutliers.count(a)) {
                if (absoluteOutliers.count(b))
                    return absoluteOutliers.at(a) < absoluteOutliers.at(b);
                return false;
            }
            if (absoluteOutliers.count(b))
                return true;
            return probabilisticOutliers.at(a) < probabilisticOutliers.at(b);
        });
        std::reverse(proposed.begin(), proposed.end());
        if (proposed.size() > maxOutliers)
            proposed.erase(proposed.begin() + maxOutliers, proposed.end());
        for (const int lmId : proposed) {
            filterState.removeLandmarkById(lmId);
            measurement.camCoordinates.erase(lmId);
        }
    }
    double getMedianSceneDepth() const { // :366-380
        const std::vector<Landmark> lms = stateEstimate().cameraLandmarks;
        std::vector<double> d2(lms.size());
        for (size_t i = 0; i < lms.size(); ++i)
            d2[i] = lms[i].p.squaredNorm();
        const auto midway = d2.begin() + d2.size() / 2;
        std::nth_element(d2.begin(), midway, d2.end());
        double median = settings.initialSceneDepth;
        if (midway != d2.end())
            median = std::pow(*midway, 0.5);
        return median;
    }
};

} // namespace orc
